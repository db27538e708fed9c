"""Where do replays of the captured forward + backward start to disagree?  Per-parameter relative difference of the
gradients between replay r and replay 0, listed from the output side of the network to the input side."""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from mm_unet_amd.loss import DICE_BCE_Loss
from test_modules_gpu import _mmnet
DEV = "cuda:0"
gen = torch.Generator().manual_seed(4)
x = torch.randn(2, 3, 64, 64, generator=gen).to(DEV)
t = (torch.rand(2, 1, 64, 64, generator=gen) > 0.88).float().to(DEV)
m = _mmnet().train()
loss_fn = DICE_BCE_Loss()
for _ in range(2):
    loss_fn(m(x), t).backward(); m.zero_grad(set_to_none=True)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = loss_fn(m(x), t); loss.backward()
names = [k for k, p in m.named_parameters()]
ref = None
worst = (0.0, None)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for r in range(R):
    g.replay(); torch.cuda.synchronize()
    snap = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    if ref is None:
        ref = snap; print("loss", float(loss)); continue
    rel = {k: float((snap[k] - ref[k]).norm() / (ref[k].norm() + 1e-30)) for k in snap}
    vals = sorted(rel.values())
    print(f"replay {r}: loss {float(loss):.6f} median {vals[len(vals)//2]:.2e} p90 {vals[int(.9*len(vals))]:.2e} max {vals[-1]:.2e}", flush=True)
    if vals[-1] > worst[0]:
        worst = (vals[-1], rel)
rel = worst[1]
print("---- worst replay, output side first (every 12th parameter + the 25 largest)")
ordered = [k for k in reversed(names) if k in rel]
for i, k in enumerate(ordered):
    if i % 12 == 0: print(f"{i:4d} {rel[k]:.2e} {k}")
for k in sorted(rel, key=rel.get, reverse=True)[:25]:
    print(f"TOP {rel[k]:.2e} idx {ordered.index(k)} {k}")
