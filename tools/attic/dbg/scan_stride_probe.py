"""Do the scan kernels lose bandwidth to power-of-two row strides?  Same data, inputs' rows L vs L + pad apart."""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from mm_unet_amd import selective_scan_hip as ss
DEV = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
b, d, l, n = 8, 128, 65536, 16
gen = torch.Generator(device=DEV).manual_seed(0)
A = -0.5 * torch.rand(d, n, device=DEV, generator=gen)
D = torch.randn(d, device=DEV, generator=gen)
bias = 0.5 * torch.rand(d, device=DEV, generator=gen)
for pad in (0, 64, 96, 1056, 0):
    def mk(rows, rand=False):
        t = torch.empty(b, rows, l + pad, device=DEV)[:, :, :l]
        t.copy_(0.5 * torch.rand(b, rows, l, device=DEV, generator=gen) if rand else torch.randn(b, rows, l, device=DEV, generator=gen))
        return t
    u, z, dout = mk(d), mk(d), mk(d)
    delta = mk(d, True)
    B = mk(n).unsqueeze(1); C = mk(n).unsqueeze(1)
    tf = timeit(lambda: ss.fwd(u, delta, A, B, C, D, z, bias, True, want_out=False))
    x = ss.fwd(u, delta, A, B, C, D, z, bias, True, want_out=False)[1]
    tb = timeit(lambda: ss.bwd(u, delta, A, B, C, D, z, bias, dout, x, None, None, True, False))
    print(f"pad {pad:5d}: fwd {tf:7.1f} us   bwd {tb:7.1f} us", flush=True)
