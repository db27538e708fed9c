"""Is gemm_nt limited by its access pattern (192 rows x 128 B at 2 MB strides per chunk) or by its pipeline?
Same arithmetic on two layouts: rows 2 MB apart (the real one) vs each chunk's tile stored contiguously."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_unet_amd import mfma_gemm
DEV = "cuda:0"
T = 8 * 65536
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (m, n) in ((256, 64), (128, 64), (128, 4)):
    a = torch.randn(m * T, device=DEV); b = torch.randn(n * T, device=DEV)
    for pad in (0, 64, 1056):
        Tp = T + pad
        ap = torch.randn(m * Tp, device=DEV); bp = torch.randn(n * Tp, device=DEV)
        for ex, nar in ((False, False), (False, True), (True, True)):
            t = timeit(lambda: mfma_gemm.gemm_nt(ap, bp, m, n, 1, T, Tp, 0, Tp, 0, exact=ex, narrow=nar))
            print(f"{m}x{n}: {'exact' if ex else 'split'} {'32' if nar else '128'}-token steps, row stride T+{pad:<5d} {t:7.1f} us  {(m + n) * T * 4 / t / 1e6:6.2f} TB/s")
    t = timeit(lambda: mfma_gemm.gemm_nt(a, b, m, n, T // 32, 32, 32, m * 32, 32, n * 32, narrow=True))
    print(f"{m}x{n}: chunk tiles contiguous  {t:7.1f} us  {(m + n) * T * 4 / t / 1e6:6.2f} TB/s")
