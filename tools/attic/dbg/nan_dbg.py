"""Graph-replay NaN hunt: MM_Net 2x3x64x64 training steps under HIP-graph replay; prints the first step whose
gradients / parameters are not finite and which tensors.  Env toggles select kernel families."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_unet_amd.loss import DICE_BCE_Loss
from mm_unet_amd.train_step import TrainStep, make_optimizer
import mm_unet_amd.mmunet as pm
DEV = "cuda:0"
gen = torch.Generator().manual_seed(4)
x = torch.randn(2, 3, 64, 64, generator=gen).to(DEV)
t = (torch.rand(2, 1, 64, 64, generator=gen) > 0.88).float().to(DEV)
torch.manual_seed(50)
m = pm.MM_Net(num_classes=1).to(DEV).train()
if os.environ.get("NODROP") == "1":
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
try:
    opt = make_optimizer(m, lr=1e-3, weight_decay=0.0, capturable=True)
except TypeError:
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.0, capturable=True)
step = TrainStep(m, DICE_BCE_Loss(), opt, use_graph=True)
for i in range(6):
    l = step(x, t)
    torch.cuda.synchronize()
    badg = [k for k, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    badp = [k for k, p in m.named_parameters() if not torch.isfinite(p).all()]
    print(os.environ.get("TAG", ""), i, float(l), "nan grads:", len(badg), badg[:4], "nan params:", len(badp), flush=True)
    if badg:
        names = [k for k, p in m.named_parameters() if p.grad is not None]
        ok = [k for k in names if k not in set(badg)]
        print("finite grads (%d):" % len(ok), ok[:80])
        mods = sorted(set(k.split(".")[0] for k in badg))
        print("top-level modules with NaN grads:", mods)
        print("top-level modules with finite grads:", sorted(set(k.split(".")[0] for k in ok)))
    if badp:
        break
