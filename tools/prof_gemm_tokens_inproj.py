"""gemm_tokens at the in_proj shape of the largest tri-directional block (256 rows x 64 inner over 8 x 65,536 tokens: four
row tiles read every X tile) -- for traffic passes (tools/pmc_traffic_generic.sh gt gemm_tokens_mfma_kernel <this>)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.mfma_gemm import gemm_tokens
M, K, T, B = 256, 64, 65536, 8
W = torch.randn(M, K, device="cuda") / 8
X = torch.randn(K, B * T, device="cuda")
out = torch.empty(B, M, T, device="cuda")
for _ in range(3):
    gemm_tokens(W, X, out, M, K, T, B, B * T, T, T, M * T)
torch.cuda.synchronize()
