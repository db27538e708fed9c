"""Every parameter gradient of the fused route (hand-written kernels, gradient hand-overs, fused loss) against the plain
ATen route of the same modules (fused_paths.plain_aten: nn.Conv2d / GroupNorm / grid_sample / interpolate ...; only the
scan and conv1d stay HIP), eager, train mode, at 512 x 512: validates the hand-overs at the shapes the benchmark runs
(vectorised max-pool / tiled resize / CBAM paths that small test inputs do not reach).  Debug aid."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_unet_amd import fused_paths, loss as loss_mod
from mm_unet_amd.mmunet import MM_Net

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = "cuda"
torch.manual_seed(50)
model = MM_Net(num_classes=1).to(dev).train()
for m in model.modules():
    if isinstance(m, torch.nn.Dropout2d):
        m.p = 0.0
ref = copy.deepcopy(model)
gen = torch.Generator().manual_seed(1)
x = torch.randn(bs, 3, size, size, generator=gen).to(dev)
t = (torch.rand(bs, 1, size, size, generator=gen) > 0.8).float().to(dev)
loss_fn = loss_mod.DICE_BCE_Loss()
la = loss_fn(model(x), t)
la.backward()
saved, loss_mod.FUSED = loss_mod.FUSED, False
with fused_paths.plain_aten():
    lb = loss_fn(ref(x), t)
    lb.backward()
loss_mod.FUSED = saved
torch.cuda.synchronize()
print(f"loss fused {float(la):.7f}  plain {float(lb):.7f}")
ga = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
gb = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
assert ga.keys() == gb.keys(), (set(ga) ^ set(gb))
total = sum(float(v.double().pow(2).sum()) for v in gb.values()) ** 0.5
rows = sorted(((float((ga[k] - gb[k]).double().norm()) / max(float(gb[k].double().norm()), 1e-30),
                float(gb[k].double().norm()) / total, k) for k in gb), reverse=True)
live = [r for r in rows if r[1] > 1e-6]
print(f"{len(ga)} gradients; overall rel diff {sum(float((ga[k] - gb[k]).double().pow(2).sum()) for k in ga) ** 0.5 / total:.3e}")
print(f"largest differences among the {len(live)} tensors with a share > 1e-6 of the gradient norm:")
for rel, share, k in live[:10]:
    print(f"  rel {rel:9.3e}  share {share:9.3e}  {k}")
bad = [r for r in live if r[0] > 0.2]
print("tensors > 20 % off:", len(bad))
sys.exit(1 if bad else 0)
