"""Per-shape time of gemm_tokens' 32-token kernel family (HIP events, 50 launches per shape) and its error against float64
-- for A/B runs over MMUNET_GEMM_TOKENS_SMALL_NB (0: default choice, 1: 32-token tiles, 4: 64 x 128 tiles on eight waves)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.mfma_gemm import gemm_tokens
dev = "cuda:0"
line = []
for (M, K, T, B) in ((512, 1536, 256, 8), (1536, 512, 256, 8), (256, 768, 1024, 8), (768, 256, 1024, 8), (128, 384, 4096, 8),
                     (384, 128, 4096, 8), (64, 384, 1024, 8)):
    W = torch.randn(M, K, device=dev) / K ** 0.5
    X = torch.randn(K, B * T, device=dev)
    out = torch.empty(B, M, T, device=dev)
    for _ in range(5):
        gemm_tokens(W, X, out, M, K, T, B, B * T, T, T, M * T)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        gemm_tokens(W, X, out, M, K, T, B, B * T, T, T, M * T)
    b.record()
    torch.cuda.synchronize()
    ref = torch.einsum("mk,kbt->bmt", W.double(), X.view(K, B, T).double())
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    line.append("%dx%dx%d: %.1f us (%.1e)" % (M, K, B * T, a.elapsed_time(b) * 20, err))
print("NB=%s  " % os.environ.get("MMUNET_GEMM_TOKENS_SMALL_NB", "0") + "  ".join(line))
