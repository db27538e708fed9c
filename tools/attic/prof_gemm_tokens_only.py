"""gemm_tokens alone (3 launches per shape) for PMC passes: tools/pmc_generic.sh gt gemm_tokens_mfma_kernel tools/prof_gemm_tokens_only.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.mfma_gemm import gemm_tokens
B = 8
for (M, K, T, nb) in [(512, 1536, 256, 8), (192, 64, 131072, 1), (64, 192, 16384, 8)]:
    W = torch.randn(M, K, device="cuda") / K ** 0.5
    X = torch.randn(K, nb * T, device="cuda")
    out = torch.empty(nb, M, T, device="cuda")
    for _ in range(3):
        gemm_tokens(W, X, out, M, K, T, nb, nb * T, T, T, M * T)
torch.cuda.synchronize()
