"""conv3x3_wgrad_mfma alone at CBAM's shape (PMC passes: tools/pmc_generic.sh wg conv3x3_wgrad_mfma_kernel <this>)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.conv3x3_mfma import _wgrad
x = torch.randn(8, 64, 256, 256, device="cuda")
g = torch.randn(8, 64, 256, 256, device="cuda")
for _ in range(3):
    _wgrad(x, g, 64)
torch.cuda.synchronize()
