"""Where do the small fill / copy / add launches of a training step come from?  One eager step under torch.profiler;
for every aten op that launches a fill-like kernel, the chain of enclosing ops (autograd nodes, python functions).
usage: python tools/dbg/fill_sources.py [pattern ...]   (default: fill_ zero_ zeros add_ copy_)"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_unet_amd.loss import DICE_BCE_Loss  # noqa: E402
from mm_unet_amd.mmunet import MM_Net  # noqa: E402
from mm_unet_amd.train_step import TrainStep, make_optimizer  # noqa: E402

pats = sys.argv[1:] or ["aten::fill_", "aten::zero_", "aten::add_", "aten::copy_", "aten::add", "aten::mul", "aten::sum", "aten::contiguous", "aten::clone"]
dev = "cuda:0"
torch.manual_seed(0)
BF16 = os.environ.get("BF16") == "1"        # BASELINE config 3: bf16 autocast, batch 16
BATCH = int(os.environ.get("BATCH", "16" if BF16 else "8"))
SIZE = int(os.environ.get("SIZE", "512"))
DSTATE = int(os.environ.get("DSTATE", "16"))
m = MM_Net(num_classes=1, d_state=DSTATE).to(dev).train()
step = TrainStep(m, DICE_BCE_Loss(), make_optimizer(m), use_graph=False, amp_dtype=torch.bfloat16 if BF16 else None)
x = torch.randn(BATCH, 3, SIZE, SIZE, device=dev)
t = (torch.rand(BATCH, 1, SIZE, SIZE, device=dev) > 0.88).float()
for _ in range(2):
    step(x, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step(x, t)
    torch.cuda.synchronize()
evs = prof.profiler.function_events
agg = collections.defaultdict(lambda: [0, 0.0])
KPAT = os.environ.get("KPAT")    # instead of op names: ops whose kernels match this substring (e.g. FillFunctor)
for e in evs:
    if not e.kernels:          # launched nothing itself
        continue
    if KPAT:
        if not any(KPAT in k.name for k in e.kernels):
            continue
    elif e.name not in pats:
        continue
    chain = []
    p = e.cpu_parent
    while p is not None and len(chain) < 4:
        chain.append(p.name)
        p = p.cpu_parent
    shape = str(e.input_shapes)[:60] if e.input_shapes else ""
    key = (e.name, " < ".join(chain)[:150], shape)
    agg[key][0] += 1
    agg[key][1] += sum(k.duration for k in e.kernels)
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = collections.Counter()
for (name, chain, shape), (n, us) in rows:
    tot[name] += n
print("launching calls per step:", dict(tot))
for (name, chain, shape), (n, us) in rows[:70]:
    print(f"{n:5d} x {name:16s} {us:9.1f} us  {shape:60s} <- {chain}")
