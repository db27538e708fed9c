"""MM_Net eval logits at 1x3x512x512: this build (matrix-core GEMM / conv paths) and the same forward with those paths on
the libraries' float32 kernels, each against the float32 AND the float64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import model_ref
import mm_unet_amd.mmunet as pm
import mm_unet_amd.conv3x3_mfma as cm
import mm_unet_amd.mfma_gemm as mg
torch.manual_seed(50)
model = pm.MM_Net(num_classes=1)
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
model = model.cuda().eval()
img = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(7))
with torch.no_grad():
    ours = model(img.cuda()).cpu()
    supported = cm.supported
    mg.ENABLED, cm.supported = False, (lambda x, w: False)
    lib = model(img.cuda()).cpu()
    mg.ENABLED, cm.supported = True, supported
    r32 = model_ref.mm_net(sd, img, training=False)
    r64 = model_ref.mm_net({k: v.double() for k, v in sd.items()}, img.double(), training=False)
d = lambda a, b: float((a.double() - b.double()).abs().max())
print(f"ours-lib {d(ours, lib):.3e}  ours-o32 {d(ours, r32):.3e}  lib-o32 {d(lib, r32):.3e}  |  vs float64: ours {d(ours, r64):.3e}  lib {d(lib, r64):.3e}  o32 {d(r32, r64):.3e}")
