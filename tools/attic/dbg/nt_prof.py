import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_unet_amd import mfma_gemm
DEV = "cuda:0"
T = 8 * 65536
for (m, n, t) in ((256, 64, T), (64, 128, T), (64, 192, T // 4)):
    g = torch.randn(m, t, device=DEV); x = torch.randn(n, t, device=DEV)
    for _ in range(5):
        mfma_gemm.gemm_nt(g, x, m, n, 1, t, t, 0, t, 0)
torch.cuda.synchronize()
