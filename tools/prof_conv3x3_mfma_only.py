"""conv3x3_mfma forward only, 3 launches per shape (PMC passes: tools/pmc_generic.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.conv3x3_mfma import conv3x3_mfma
shapes = [(8, 64, 64, 256, 256), (8, 256, 256, 64, 64)]
for (B, Cin, Cout, H, W) in shapes:
    x = torch.randn(B, Cin, H, W, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        conv3x3_mfma(x, w, None)
torch.cuda.synchronize()
