import sys, torch
sys.path.insert(0, ".")
from mm_unet_amd.mfma_gemm import gemm_tokens
dev = "cuda:0"
for (M, K, T, B) in ((512, 1536, 256, 8), (128, 384, 4096, 8)):
    W = torch.randn(M, K, device=dev) / K ** 0.5
    X = torch.randn(K, B * T, device=dev)
    out = torch.empty(B, M, T, device=dev)
    for _ in range(10):
        gemm_tokens(W, X, out, M, K, T, B, B * T, T, T, M * T)
    torch.cuda.synchronize()
