"""Which zero fills / adds / copies does ATen run in one training step (float32, 8 x 3 x 512 x 512)?  Shapes and, where
the call comes from Python code of the package, the call site."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from mm_unet_amd.mmunet import MM_Net
from mm_unet_amd.loss import DICE_BCE_Loss

seen = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("aten.zeros", "aten.zero_", "aten.fill_", "aten.zeros_like", "aten.add.Tensor", "aten.add_.Tensor", "aten.copy_", "aten.clone", "aten.new_zeros", "aten.full")):
            shapes = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))
            if not shapes and args and isinstance(args[0], (list, tuple)):
                shapes = (tuple(args[0]),)
            site = "(autograd engine)"
            for fr in reversed(traceback.extract_stack()):
                if "mm-unet_amd" in fr.filename or "mm_unet_amd" in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            seen[(name, shapes, site)] += 1
        return func(*args, **(kwargs or {}))

torch.manual_seed(50)
m = MM_Net(num_classes=1).cuda().train()
x = torch.randn(8, 3, 512, 512, device="cuda"); t = (torch.rand(8, 1, 512, 512, device="cuda") > 0.88).float()
DICE_BCE_Loss()(m(x), t).backward()
m.zero_grad(set_to_none=True)
with Log():
    DICE_BCE_Loss()(m(x), t).backward()
tot = collections.Counter()
for (name, shapes, site), n in seen.items():
    tot[name] += n
print(dict(tot))
for (name, shapes, site), n in sorted(seen.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n:3d} x {name:26s} {site:34s} {shapes}")
