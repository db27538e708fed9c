"""Times the projection GEMMs around the scan in the reference's formulation vs the transposed
([R][B*L]) formulation, at the RCG shapes (D=128, R=36, r=4)."""
import torch, sys
DEV = "cuda:0"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (b, d, l, R, r) in [(8, 128, 65536, 36, 4), (8, 128, 16384, 36, 4), (8, 6, 16384, 33, 1)]:
    BL = b * l
    conv_m = torch.randn(d, BL, device=DEV)          # [D][B*L]  (the layout conv1d_out has)
    Wx = torch.randn(R, d, device=DEV)
    Wdt = torch.randn(d, r, device=DEV)
    dxT = torch.randn(R, BL, device=DEV)
    dx = dxT.t().contiguous()                         # [BL, R]
    ddelta = torch.randn(d, BL, device=DEV)
    print(f"--- B{b} D{d} L{l}")
    print("ref  x_dbl = linear(conv_m.t(), Wx)          %8.1f us" % timeit(lambda: torch.nn.functional.linear(conv_m.t(), Wx)))
    print("new  x_dblT = Wx @ conv_m                    %8.1f us" % timeit(lambda: Wx @ conv_m))
    x_dbl = torch.nn.functional.linear(conv_m.t(), Wx); xT = Wx @ conv_m
    print("ref  delta = Wdt @ x_dbl[:, :r].t()          %8.1f us" % timeit(lambda: Wdt @ x_dbl[:, :r].t()))
    print("new  delta = Wdt @ x_dblT[:r]                %8.1f us" % timeit(lambda: Wdt @ xT[:r]))
    print("ref  B = x_dbl[:, r:r+16] -> (b n l) contig  %8.1f us" % timeit(lambda: x_dbl[:, r:r+16].reshape(b, l, 16).permute(0, 2, 1).contiguous()))
    print("ref  dWx = dx.t() @ conv_m.t()               %8.1f us" % timeit(lambda: dx.t() @ conv_m.t()))
    print("new  dWx = dxT @ conv_m.t()                  %8.1f us" % timeit(lambda: dxT @ conv_m.t()))
    print("ref  dconv += Wx.t() @ dx.t()  (addmm)       %8.1f us" % timeit(lambda: torch.addmm(ddelta, Wx.t(), dx.t())))
    print("new  dconv += Wx.t() @ dxT     (addmm)       %8.1f us" % timeit(lambda: torch.addmm(ddelta, Wx.t(), dxT)))
    print("ref  dWdt = ddelta @ x_dbl[:, :r]            %8.1f us" % timeit(lambda: ddelta @ x_dbl[:, :r]))
    print("new  dWdt = ddelta @ x_dblT[:r].t()          %8.1f us" % timeit(lambda: ddelta @ xT[:r].t()))
    print("ref  dx[:, :r] = ddelta.t() @ Wdt            %8.1f us" % timeit(lambda: ddelta.t() @ Wdt))
    print("new  dxT[:r] = Wdt.t() @ ddelta              %8.1f us" % timeit(lambda: Wdt.t() @ ddelta))
    print("     copy of a [D, BL] tensor (floor)        %8.1f us" % timeit(lambda: conv_m.clone()))
