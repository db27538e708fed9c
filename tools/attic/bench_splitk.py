import torch
DEV = "cuda:0"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def splitk(X, Y, ts):
    # X [I, T], Y [J, T] -> X @ Y^T [I, J] with K = T split into T/ts batches
    I, T = X.shape; J = Y.shape[0]; S = T // ts
    return torch.bmm(X.view(I, S, ts).transpose(0, 1), Y.view(J, S, ts).permute(1, 2, 0)).sum(0)
for (b, d, l, R, r) in [(8, 128, 65536, 36, 4), (8, 128, 16384, 36, 4), (8, 6, 65536, 33, 1), (8, 6, 16384, 33, 1), (8, 6, 1024, 33, 1)]:
    T = b * l
    X = torch.randn(d, T, device=DEV); Yr = torch.randn(r, T, device=DEV); YR = torch.randn(R, T, device=DEV)
    ref1 = X @ Yr.t(); ref2 = YR @ X.t()
    print(f"--- B{b} D{d} L{l}:  plain dWdt {timeit(lambda: X @ Yr.t()):8.1f} us   plain dWx {timeit(lambda: YR @ X.t()):8.1f} us")
    for ts in (512, 2048, 8192):
        if T % ts: continue
        o1 = splitk(X, Yr, ts); o2 = splitk(YR, X, ts)
        e1 = float((o1 - ref1).abs().max() / ref1.abs().max()); e2 = float((o2 - ref2).abs().max() / ref2.abs().max())
        print(f"    split-K ts={ts:5d}: dWdt {timeit(lambda: splitk(X, Yr, ts)):8.1f} us  dWx {timeit(lambda: splitk(YR, X, ts)):8.1f} us   relerr {e1:.1e} {e2:.1e}")
