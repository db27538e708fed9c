"""Micro-benchmarks of the hand-written kernels at MM-UNet's census shapes.
Run on the GPU box:  python tools/bench_kernels.py [--bf16]
Prints per-call time and algorithmic GB/s (bytes: SURVEY.md 8d)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd import causal_conv1d_hip as cc, selective_scan_hip as ss  # noqa: E402

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


def main():
    dt = torch.bfloat16 if "--bf16" in sys.argv else torch.float32
    s = 2 if dt == torch.bfloat16 else 4
    print(f"dtype {dt}")
    for (b, d, l, n) in [(8, 128, 65536, 16), (8, 128, 16384, 16), (8, 128, 4096, 16), (8, 6, 65536, 16),
                         (8, 6, 16384, 16), (8, 6, 4096, 16), (8, 6, 1024, 16), (8, 6, 256, 16), (8, 2, 4096, 16)]:
        g = torch.Generator(device=DEV).manual_seed(0)
        A = -0.5 * torch.rand(d, n, device=DEV, generator=g)
        B = torch.randn(b, 1, n, l, device=DEV, generator=g).to(dt)
        C = torch.randn(b, 1, n, l, device=DEV, generator=g).to(dt)
        D = torch.randn(d, device=DEV, generator=g)
        bias = 0.5 * torch.rand(d, device=DEV, generator=g)
        mk = lambda: torch.randn(d, b, l, device=DEV, generator=g).to(dt).permute(1, 0, 2)  # noqa: E731
        u, z, dout = mk(), mk(), mk()
        delta = (0.5 * torch.rand(d, b, l, device=DEV, generator=g)).to(dt).permute(1, 0, 2)
        res = ss.fwd(u, delta, A, B, C, D, z, bias, True)
        tf = timeit(lambda: ss.fwd(u, delta, A, B, C, D, z, bias, True, want_out=False))
        tb = timeit(lambda: ss.bwd(u, delta, A, B, C, D, z, bias, dout, res[1], None, None, True, False))
        bf = s * b * l * (4 * d + 2 * n)
        bb = s * b * l * (8 * d + 2 * n) + 4 * b * l * 2 * n
        print(f"scan B{b} D{d:<3} L{l:<6} N{n}: fwd {tf*1e3:8.1f} us {bf/tf/1e6:7.0f} GB/s | "
              f"bwd {tb*1e3:8.1f} us {bb/tb/1e6:7.0f} GB/s")
        w = torch.randn(d, 4, device=DEV, generator=g)
        cb = torch.randn(d, device=DEV, generator=g)
        t1 = timeit(lambda: cc.causal_conv1d_fwd(u, w, cb, True))
        t2 = timeit(lambda: cc.causal_conv1d_bwd(u, w, cb, dout, None, True))
        print(f"conv B{b} D{d:<3} L{l:<6}    : fwd {t1*1e3:8.1f} us {2*s*b*d*l/t1/1e6:7.0f} GB/s | "
              f"bwd {t2*1e3:8.1f} us {3*s*b*d*l/t2/1e6:7.0f} GB/s")


if __name__ == "__main__":
    main()
