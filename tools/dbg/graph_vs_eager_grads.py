"""Every parameter gradient of the captured training step against the eager backward pass, at the benchmark's size
(3x512x512, bs 8 by default): after two warm-ups + capture, a REPLAY on a fresh batch must leave in p.grad what an eager
backward on that batch produces.  A gradient read before its deferred sum, a stale input buffer or a hand-over that
depends on capture-time state shows as an O(1) difference on that tensor (run-to-run noise of the float atomics: < 1 %
on all but noise-level tensors).  Debug aid; tests/test_modules_gpu.py::test_train_step_graph_replay_matches_eager is the
unit-test form at 64 / 256.  (Third argument ``bf16``: under autocast.  Not conclusive: two EAGER bf16 runs of the same
batch already differ by ~55 % in the gradient norm at 512 x 512 -- the sampler's atomics flip bf16 roundings and the
train-mode network amplifies them -- so only the float32 form can show a wrong-step gradient.)"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_unet_amd.mmunet import MM_Net
from mm_unet_amd.loss import DICE_BCE_Loss
from mm_unet_amd.train_step import TrainStep, make_optimizer

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
amp = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else None     # BASELINE config 3's autocast
dev = "cuda"
torch.manual_seed(50)
model = MM_Net(num_classes=1).to(dev).train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout2d):
        m.p = 0.0
ref = copy.deepcopy(model)
gen = torch.Generator().manual_seed(1)
batches = [(torch.randn(bs, 3, size, size, generator=gen).to(dev), (torch.rand(bs, 1, size, size, generator=gen) > 0.8).float().to(dev))
           for _ in range(4)]
step = TrainStep(model, DICE_BCE_Loss(), make_optimizer(model, lr=0.0, capturable=True), use_graph=True, amp_dtype=amp)
for x, t in batches:
    step(x, t)
torch.cuda.synchronize()
assert step._graph is not None
x, t = batches[-1]
loss_fn = DICE_BCE_Loss()
ref.zero_grad(set_to_none=True)
if amp is not None:
    with torch.autocast("cuda", dtype=amp):
        logits = ref(x)
    loss_fn(logits.float(), t).backward()
else:
    loss_fn(ref(x), t).backward()
torch.cuda.synchronize()
gr = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
gg = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
assert gr.keys() == gg.keys(), (set(gr) ^ set(gg))
total = sum(float(v.double().pow(2).sum()) for v in gr.values()) ** 0.5
rows = []
for k in gr:
    n = float(gr[k].double().norm())
    rel = float((gr[k] - gg[k]).double().norm()) / max(n, 1e-30)
    rows.append((rel, n / total, k))
rows.sort(reverse=True)
print(f"{len(gr)} gradients; overall rel diff {sum(float((gr[k] - gg[k]).double().pow(2).sum()) for k in gr) ** 0.5 / total:.3e}")
live = [r for r in rows if r[1] > 1e-9]        # (below: analytically zero gradients -- a GroupNorm bias under a BatchNorm)
print(f"largest differences among the {len(live)} tensors with a share > 1e-9 of the gradient norm:")
for rel, share, k in live[:10]:
    print(f"  rel {rel:9.3e}  share {share:9.3e}  {k}")
limit = 0.05 if amp is None else 0.5      # (bf16: rounding noise of a few %; a gradient of the wrong step is ~140 % off)
bad = [r for r in live if r[0] > limit and (amp is None or r[1] > 1e-4)]
print(f"tensors > {limit:.0%} off:", len(bad))
sys.exit(1 if bad else 0)
