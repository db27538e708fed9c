"""model.py's plain Unet (every layer a dense 3x3 conv, 64-1024 channels), training step
fwd + Dice+BCE + bwd on 8 x 3 x 512 x 512, with the matrix-core convolutions (csrc/conv3x3_mfma.hip: forward and input
gradient; csrc/conv3x3_wgrad_mfma.hip: weight gradient) and with MIOpen only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mm_unet_amd.conv3x3_mfma as cm
from mm_unet_amd.loss import DICE_BCE_Loss
from mm_unet_amd.unet import Unet
dev = "cuda:0"
torch.manual_seed(50)
model = Unet(3, 1).to(dev).train()
x = torch.randn(8, 3, 512, 512, device=dev)
t = (torch.rand(8, 1, 512, 512, device=dev) > 0.88).float()
loss_fn = DICE_BCE_Loss()
supported = cm.supported
def run(n):
    for _ in range(n):
        model.zero_grad(set_to_none=True)
        loss_fn(model(x), t).backward()
    torch.cuda.synchronize()
for name, on in (("matrix-core convolutions (forward, input gradient, weight gradient)", True), ("MIOpen only", False)):
    cm.supported = supported if on else (lambda x_, w_: False)
    run(3)
    t0 = time.perf_counter()
    run(10)
    dt = (time.perf_counter() - t0) / 10
    with torch.no_grad():
        model.eval()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            model(x)
        torch.cuda.synchronize()
        ti = (time.perf_counter() - t1) / 10
        model.train()
    print(f"{name}: train step {dt*1e3:.1f} ms = {8/dt:.1f} img/s; eval forward {ti*1e3:.1f} ms = {8/ti:.1f} img/s", flush=True)
cm.supported = supported
