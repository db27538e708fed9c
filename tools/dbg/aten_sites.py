"""Which call sites still launch plain ATen kernels (fills, copies, adds, cats, ...) in one training step
(float32, 8 x 3 x 512 x 512; DT=bf16: under bf16 autocast, batch 16)?  Logs every aten op outside a small allow-list with its shapes and the innermost
mm-unet_amd frame (autograd-engine calls have none: '?')."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from mm_unet_amd.mmunet import MM_Net
from mm_unet_amd.loss import DICE_BCE_Loss

SKIP = ("aten.view", "aten._unsafe_view", "aten.reshape", "aten.permute", "aten.transpose", "aten.t.", "aten.slice", "aten.select",
        "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.as_strided", "aten.empty", "aten.detach", "aten.alias",
        "aten.split", "aten.chunk", "aten.unbind", "aten.stride", "aten.size", "aten.is_", "aten.sym_", "aten.lift",
        "aten._local_scalar", "aten.new_empty", "aten.empty_like", "aten.empty_strided", "aten.narrow", "aten.unfold")
seen = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(name.startswith(k) for k in SKIP):
            shapes = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))[:3]
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if ("mm-unet_amd" in fr.filename or "mm_unet_amd" in fr.filename):
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            seen[(name, shapes, site)] += 1
        return func(*args, **(kwargs or {}))

torch.manual_seed(50)
BF = os.environ.get("DT") == "bf16"
NB = 16 if BF else 8
m = MM_Net(num_classes=1).cuda().train()
x = torch.randn(NB, 3, 512, 512, device="cuda"); t = (torch.rand(NB, 1, 512, 512, device="cuda") > 0.88).float()
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=BF):
        out = m(x)
    DICE_BCE_Loss()(out.float(), t).backward()   # (the loss outside autocast, as train_step.py runs it)
step()   # warm
m.zero_grad(set_to_none=True)
with Log():
    step()
agg = collections.Counter()
for (name, shapes, site), n in seen.items():
    agg[(name, site)] += n
for (name, site), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    ex = next(s for (nm, s, st), _ in seen.items() if nm == name and st == site)
    print(f"{n:4d} x {name:40s} {site:34s} e.g. {ex}")
