"""Device-to-device copy and read-only / write-only streams: what the HBM roofline looks like on this box
(SURVEY.md 8d asks to confirm the 8 TB/s peak with a copy benchmark)."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 28          # 1 GiB of float32
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
t = timed(lambda: y.copy_(x))
print(f"copy 1 GiB -> 1 GiB: {t*1e3:.3f} ms = {2 * n * 4 / t / 1e12:.2f} TB/s (read + write)")
t = timed(lambda: x.sum())
print(f"read-only (sum of 1 GiB): {t*1e3:.3f} ms = {n * 4 / t / 1e12:.2f} TB/s")
t = timed(lambda: y.fill_(1.0))
print(f"write-only (fill 1 GiB): {t*1e3:.3f} ms = {n * 4 / t / 1e12:.2f} TB/s")
t = timed(lambda: torch.add(x, 1.0, out=y))
print(f"elementwise x + 1 -> y: {t*1e3:.3f} ms = {2 * n * 4 / t / 1e12:.2f} TB/s (read + write)")
