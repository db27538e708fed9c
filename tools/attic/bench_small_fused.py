"""Times csrc/mamba_small_fused.hip directly (no autograd, no graph): forward / backward per call for a few map sizes
and state counts; the state count separates the fixed cost (staging, pre-phase, epilogue) from the per-state cost of
the scan loop.  usage: python tools/bench_small_fused.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm_unet_amd import _lib

dev = "cuda:0"
L = _lib.lib()


def run(B, K, H, W, N, parts=None, iters=200):
    D = 2 * K
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    off = torch.tanh(r(B, 2 * K, H, W))
    w_in, cw, cb, wx, wdt, dtb = r(4 * K, K) * 0.5, r(D, 4) * 0.5, r(D) * 0.1, r(1 + 2 * N, D) * 0.3, r(D) * 0.3, r(D) * 0.1
    A = -torch.exp(torch.log(torch.arange(1, N + 1, dtype=torch.float32)).repeat(D, 1)).to(dev).contiguous()
    Dp, wout, al = torch.ones(D, device=dev), r(K, D) * 0.3, torch.tensor([0.54], device=dev)
    parts = parts or L.mmu_mamba_small_parts(B, K, H, W, N, 0)
    y = torch.empty(parts, B, K, H, W, device=dev)
    dy = r(B, K, H, W)
    doff = torch.empty_like(off)
    nv = L.mmu_mamba_small_grad_floats(K, N)
    ws = torch.empty(L.mmu_mamba_small_bwd_workspace_floats(B, K, H, W, N, parts), device=dev)
    dw = torch.empty(nv, device=dev)
    p = _lib.MambaSmallParams()
    p.batch, p.height, p.width, p.taps, p.dstate, p.parts, p.extend_scope = B, H, W, K, N, parts, 1.0
    p.offset, p.in_proj_weight, p.conv_weight, p.conv_bias = off.data_ptr(), w_in.data_ptr(), cw.data_ptr(), cb.data_ptr()
    p.x_proj_weight, p.dt_proj_weight, p.dt_bias, p.A, p.D = wx.data_ptr(), wdt.data_ptr(), dtb.data_ptr(), A.data_ptr(), Dp.data_ptr()
    p.out_proj_weight, p.altho, p.y = wout.data_ptr(), al.data_ptr(), y.data_ptr()
    p.dy, p.doffset, p.workspace, p.dweights = dy.data_ptr(), doff.data_ptr(), ws.data_ptr(), dw.data_ptr()
    st = _lib.stream_of(off)
    res = []
    for fn in (L.mmu_mamba_small_fwd, L.mmu_mamba_small_bwd):
        for _ in range(5):
            _lib.check(fn(p, st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn(p, st)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters * 1e3)
    return res


if __name__ == "__main__":
    for (B, K, H, W) in ((8, 3, 16, 16), (8, 3, 32, 32), (8, 1, 32, 32)):
        for N, parts in ((16, 2), (16, 4), (16, 8), (16, 16), (64, 8), (64, 16)):
            f, b = run(B, K, H, W, N, parts)
            print(f"B {B} K {K} {H}x{W} N {N:3d} parts {parts:2d}: fwd {f:7.1f} us   bwd(+reduce) {b:7.1f} us", flush=True)
