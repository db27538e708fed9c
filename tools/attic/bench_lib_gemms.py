"""How far from the HBM roofline are the library GEMMs the training step still uses?  The token-contraction weight
gradients of the 128-channel Mamba blocks (B = 8, L = 65,536: 524,288 tokens), float32: as one library GEMM, and as
the framework runs them (tall_gemm.nt_splitk: slabs of the token axis as a batched GEMM + an ordered sum).
Prints time, operand bytes / time and FLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.tall_gemm import nt_splitk  # noqa: E402
from mm_unet_amd import mfma_gemm  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


T = 8 * 65536
cases = [
    ("in_proj  dW  (256 x T) . (T x 64)", 256, 64, T),
    ("out_proj dW  (64 x T) . (T x 128)", 64, 128, T),
    ("x_proj   dW  (36 x T) . (T x 128)", 36, 128, T),
    ("dt_proj  dW  (128 x T) . (T x 4)", 128, 4, T),
    ("DSC      dW  (64 x T/4) . (T/4 x 192)", 64, 192, T // 4),
]
for name, m, n, t in cases:
    g = torch.randn(m, t, device=DEV)
    x = torch.randn(n, t, device=DEV)
    byt = 4.0 * t * (m + n)
    def batched():
        mfma_gemm.NT_ENABLED = False          # the earlier form: slabs of the token axis as one batched library GEMM + a sum
        try:
            return nt_splitk(g, x)
        finally:
            mfma_gemm.NT_ENABLED = True
    for how, fn in (("one GEMM", lambda: g @ x.t()), ("batched slabs", batched),
                    ("gemm_nt HIP", lambda: mfma_gemm.gemm_nt(g, x, m, n, 1, t, t, 0, t, 0)),
                    ("gemm_nt exact", lambda: mfma_gemm.gemm_nt(g, x, m, n, 1, t, t, 0, t, 0, exact=True))):
        ms = timeit(fn)
        print(f"{name:42s} {how:10s} {ms*1e3:8.1f} us  {byt/ms/1e6:7.0f} GB/s of operands  {2.0*m*n*t/ms/1e9:7.1f} TFLOP/s")
