"""CBAM (src/UM_Net/MMUNet.py:313-338) forward + backward on the stem map [8, 64, 256, 256] (run under tools/kstats.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.mmunet import CBAM
torch.manual_seed(0)
m = CBAM(64).cuda()
x = torch.relu(torch.randn(8, 64, 256, 256, device="cuda")).requires_grad_()
g = torch.randn(8, 64, 256, 256, device="cuda")
for _ in range(5):
    m(x).backward(g)
torch.cuda.synchronize()
