import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.mfma_gemm import gemm_tokens
for (M, K, T, B) in ((256, 64, 65536, 8), (256, 64, 16384, 8), (128, 64, 65536, 8), (64, 192, 65536, 8)):
    W = torch.randn(M, K, device="cuda") / 8
    X = torch.randn(K, B * T, device="cuda")
    out = torch.empty(B, M, T, device="cuda")
    for _ in range(3):
        gemm_tokens(W, X, out, M, K, T, B, B * T, T, T, M * T)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        gemm_tokens(W, X, out, M, K, T, B, B * T, T, T, M * T)
    b.record(); torch.cuda.synchronize()
    print("%dx%dx%d: %.1f us" % (M, K, B * T, a.elapsed_time(b) * 50), end="  ")
print()
