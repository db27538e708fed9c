"""Which python lines of this package cause the copy / add / flip / fill GPU time of one training step."""
import os, sys, torch, collections
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(x, t)
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::add", "aten::add_", "aten::flip", "aten::fill_", "aten::zero_", "aten::sum", "aten::mul", "aten::cat", "aten::clone", "aten::contiguous")
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.key_averages(group_by_stack_n=12):
    if e.key not in want or e.self_device_time_total < 50:
        continue
    site = next((s for s in e.stack if "mm-unet_amd" in s or "mm_unet_amd" in s), None)
    if site is None:
        site = "(autograd engine / other) " + (e.stack[0] if e.stack else "")
    k = (e.key, site.split("/")[-1][:90])
    agg[k][0] += e.self_device_time_total
    agg[k][1] += e.count
for (op, site), (tt, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{tt/1e3:7.2f} ms n={n:4d} {op:16s} {site}")
