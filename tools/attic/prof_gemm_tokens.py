"""DSC-conv GEMM shapes: matrix-core gemm_tokens vs the hipBLASLt path (run under tools/kstats.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mm_unet_amd.mfma_gemm as mg
from mm_unet_amd.tall_gemm import dsc_gemm
B = 8
for (Cin, Cout, T) in [(64, 64, 128 * 128), (128, 128, 64 * 64), (256, 256, 32 * 32), (512, 512, 16 * 16)]:
    W2 = (torch.randn(Cout, Cin * 3, device="cuda") / (Cin * 3) ** 0.5).requires_grad_()
    S = torch.randn(Cin * 3, B * T, device="cuda", requires_grad=True)
    g = torch.randn(B, Cout, T, device="cuda")
    mg.MIN_TILES = 1
    for on in (True, False):
        mg.ENABLED = on
        for _ in range(5):
            dsc_gemm(W2, S, B).backward(g)
torch.cuda.synchronize()
