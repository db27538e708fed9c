"""Device time of the convolutions of one MM_Net training step, grouped by input/weight shape."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(x, t)
    torch.cuda.synchronize()
want = ("aten::miopen_convolution", "aten::convolution_backward", "aten::miopen_convolution_transpose",
        "aten::native_group_norm", "aten::native_group_norm_backward", "aten::miopen_batch_norm",
        "aten::miopen_batch_norm_backward", "aten::upsample_bilinear2d", "aten::upsample_bilinear2d_backward",
        "aten::mm", "aten::bmm", "aten::addmm")
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in want]
tot = {}
for e in rows:
    tot[e.key] = tot.get(e.key, 0) + e.device_time_total
print({k: round(v / 1e3, 2) for k, v in tot.items()})
for e in sorted(rows, key=lambda e: -e.device_time_total)[:60]:
    print(f"{e.device_time_total/1e3:7.2f} ms n={e.count:3d} {e.key.replace('aten::',''):28s} {str(e.input_shapes)[:110]}")
