"""torch.profiler view of one steady-state MM_Net training step: GPU time by aten op."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.loss import DICE_BCE_Loss
from mm_unet_amd.mmunet import MM_Net
from mm_unet_amd.train_step import TrainStep, make_optimizer
dev = torch.device("cuda", 0)
torch.manual_seed(50)
model = MM_Net(num_classes=1).to(dev).train()
step = TrainStep(model, DICE_BCE_Loss(), make_optimizer(model))
g = torch.Generator(device=dev).manual_seed(1000)
x = torch.randn(8, 3, 512, 512, device=dev, generator=g)
t = (torch.rand(8, 1, 512, 512, device=dev, generator=g) > 0.88).float()
for _ in range(3):
    step(x, t)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(x, t)
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in ka)
print(f"total self device time {tot/1e3:.1f} ms")
for e in rows[:45]:
    print(f"{e.self_device_time_total/1e3:8.2f} ms  n={e.count:5d}  cpu {e.self_cpu_time_total/1e3:7.2f} ms  {e.key[:90]}")
print("total CPU self time ms:", sum(e.self_cpu_time_total for e in ka) / 1e3)
