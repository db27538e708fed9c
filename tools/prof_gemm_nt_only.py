"""Three launches of the token-contraction kernel at bench.py's roofline_gemm_nt shape (for the PMC traffic passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd import mfma_gemm
dev = "cuda:0"
m, n, b, l = 128, 64, 8, 65536
ga = torch.randn(m, b, l, device=dev); gb = torch.randn(n, b, l, device=dev)
for _ in range(3):
    mfma_gemm.gemm_nt(ga, gb, m, n, b, l, b * l, l, b * l, l)
torch.cuda.synchronize()
