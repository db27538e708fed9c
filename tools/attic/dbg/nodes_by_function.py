"""Launches per training step grouped by the autograd Function / module call that issues them (eager step under
torch.profiler): where does the launch chain come from?"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_unet_amd.loss import DICE_BCE_Loss  # noqa: E402
from mm_unet_amd.mmunet import MM_Net  # noqa: E402
from mm_unet_amd.train_step import TrainStep, make_optimizer  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
m = MM_Net(num_classes=1).to(dev).train()
step = TrainStep(m, DICE_BCE_Loss(), make_optimizer(m), use_graph=False)
x = torch.randn(8, 3, 512, 512, device=dev)
t = (torch.rand(8, 1, 512, 512, device=dev) > 0.88).float()
for _ in range(2):
    step(x, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(x, t)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0])
seen_roots = collections.Counter()
for e in prof.profiler.function_events:
    if not e.kernels:
        continue
    p, root = e, e
    while p.cpu_parent is not None:
        p = p.cpu_parent
        if "evaluate_function" in p.name:
            continue
        root = p
    name = root.name.replace("autograd::engine::evaluate_function: ", "")
    agg[name][0] += len(e.kernels)
    agg[name][1] += sum(k.duration for k in e.kernels)
for e in prof.profiler.function_events:
    if e.cpu_parent is None or ("evaluate_function" in (e.cpu_parent.name if e.cpu_parent else "") and e.cpu_parent.cpu_parent is None):
        seen_roots[e.name.replace("autograd::engine::evaluate_function: ", "")] += 1
tot = sum(v[0] for v in agg.values())
print("launches per step:", tot)
for name, (n, us, _) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    calls = max(seen_roots.get(name, 1), 1)
    print(f"{n:5d} launches  {us/1e3:7.2f} ms  ~{n/calls:5.1f} per call x {calls:4d}  {name[:90]}")
