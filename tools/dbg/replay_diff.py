"""Do two replays of the captured forward+backward (same weights, same input, no dropout, no optimizer) give the same
gradients?  Lists the parameters whose gradients differ between replay 1 and replay 2/3, in module order."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_unet_amd.loss import DICE_BCE_Loss
import mm_unet_amd.mmunet as pm
DEV = "cuda:0"
S = int(os.environ.get("SIZE", "64")); B = int(os.environ.get("BATCH", "2"))
gen = torch.Generator().manual_seed(4)
x = torch.randn(B, 3, S, S, generator=gen).to(DEV)
t = (torch.rand(B, 1, S, S, generator=gen) > 0.88).float().to(DEV)
torch.manual_seed(50)
m = pm.MM_Net(num_classes=1).to(DEV).train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout2d):
        mod.p = 0.0
loss_fn = DICE_BCE_Loss()
for _ in range(2):
    loss_fn(m(x), t).backward()
    m.zero_grad(set_to_none=True)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = loss_fn(m(x), t)
    loss.backward()
snaps = []
for r in range(3):
    g.replay()
    torch.cuda.synchronize()
    snaps.append({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    print("replay", r, "loss", float(loss), "non-finite grads:", sum(int(not torch.isfinite(v).all()) for v in snaps[-1].values()))
for r in (1, 2):
    diff = [(k, float((snaps[r][k] - snaps[0][k]).abs().max()), float(snaps[0][k].abs().max())) for k in snaps[0]
            if not torch.equal(snaps[r][k], snaps[0][k])]
    print(f"replay {r} vs 0: {len(diff)} of {len(snaps[0])} gradients differ")
    same = [k for k in snaps[0] if torch.equal(snaps[r][k], snaps[0][k])]
    print("   identical:", same)
    small = [(k, d, s) for k, d, s in diff if d == d and d < 1e-2 * max(s, 1e-12)]
    print("   differ slightly (<1%):", [k for k, _, _ in small][:40])
    big = [(k, d, s) for k, d, s in diff if not (d == d and d < 1e-2 * max(s, 1e-12))]
    print("   differ grossly:", len(big), [k for k, _, _ in big][-30:])
