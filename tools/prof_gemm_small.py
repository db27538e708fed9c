"""The deep-inner / few-token products of one training step (library GEMMs until round 4) on gemm_tokens' 32-token
kernel against torch.bmm, under tools/kstats.sh: python3 tools/prof_gemm_small.py [reps]"""
import sys
import torch
sys.path.insert(0, ".")
from mm_unet_amd.mfma_gemm import gemm_tokens

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
for (M, K, T, B) in ((512, 1536, 256, 8), (1536, 512, 256, 8), (256, 768, 1024, 8), (768, 256, 1024, 8), (128, 384, 4096, 8),
                     (64, 128, 4096, 8), (256, 64, 4096, 8), (16, 192, 1024, 8), (64, 384, 1024, 8)):
    W = torch.randn(M, K, device=dev) / K ** 0.5
    X = torch.randn(K, B * T, device=dev)
    out = torch.empty(B, M, T, device=dev)
    Xb = X.view(K, B, T).permute(1, 0, 2)
    for _ in range(reps):
        gemm_tokens(W, X, out, M, K, T, B, B * T, T, T, M * T)
        ref = torch.bmm(W.unsqueeze(0).expand(B, M, K), Xb)
    torch.cuda.synchronize()
    print(M, K, T, B, float((out - ref).abs().max()))
