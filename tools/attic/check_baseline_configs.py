import sys, torch, time
sys.path.insert(0, "/root/repo")
from mm_unet_amd.mmunet import MM_Net
from mm_unet_amd.unet import Unet
from mm_unet_amd.loss import DICE_BCE_Loss
dev = "cuda"
def run(name, model, x, amp=None, train=True):
    t = (torch.rand(x.shape[0], 1, *x.shape[2:], device=dev) > 0.88).float()
    model = model.to(dev).train(train)
    for i in range(2):
        t0 = time.time()
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            out = model(x)
        if train:
            loss = DICE_BCE_Loss()(out.float(), t)
            loss.backward()
        torch.cuda.synchronize()
    print(f"{name}: out {tuple(out.shape)} {out.dtype} finite={bool(torch.isfinite(out).all())} {time.time()-t0:.2f}s", flush=True)
torch.manual_seed(0)
run("C2 inference bs8 512 fp32", MM_Net(num_classes=1), torch.randn(8, 3, 512, 512, device=dev), train=False)
run("C3 train bs16 512 bf16", MM_Net(num_classes=1), torch.randn(16, 3, 512, 512, device=dev), amp=torch.bfloat16)
run("C5 train bs1 1024 bf16 d_state64", MM_Net(num_classes=1, d_state=64), torch.randn(1, 3, 1024, 1024, device=dev), amp=torch.bfloat16)
run("C1 Unet bs2 256 fp32", Unet(3, 1), torch.randn(2, 3, 256, 256, device=dev))
