import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_unet_amd import selective_scan_hip as ss
DEV = "cuda:0"
def run(b, d, l, g, has_z, want_out=True):
    n = 16
    gen = torch.Generator().manual_seed(21)
    A = -0.5 * torch.rand(d, n, generator=gen).to(DEV)
    B = torch.randn(b, g, n, l, generator=gen).to(DEV); C = torch.randn(b, g, n, l, generator=gen).to(DEV)
    D = torch.randn(d, generator=gen).to(DEV); bias = (0.5 * torch.rand(d, generator=gen)).to(DEV)
    u = torch.randn(b, d, l, generator=gen).to(DEV); z = torch.randn(b, d, l, generator=gen).to(DEV) if has_z else None
    delta = (0.5 * torch.rand(b, d, l, generator=gen)).to(DEV)
    res = ss.fwd(u, delta, A, B, C, D, z, bias, True, want_out=want_out)
    os.environ["MMU_SCAN_STREAM"] = "0"
    ref = ss.fwd(u, delta, A, B, C, D, z, bias, True, want_out=want_out)
    del os.environ["MMU_SCAN_STREAM"]
    torch.cuda.synchronize()
    for i, nm in ((0, "out"), (2, "out_z")):
        if i < len(res) and res[i] is not None:
            e = (res[i] - ref[i]).abs()
            idx = torch.nonzero(e > 1e-2)
            print(f"b{b} d{d} l{l} g{g} z{has_z}: {nm} max err {e.max().item():.3e}  nbad {idx.shape[0]}", idx[:5].tolist(), idx[-3:].tolist())
    e = (res[1][..., 1::2] - ref[1][..., 1::2]).abs()
    print("   states err", e.max().item())
run(2, 256, 512, 2, False)
run(2, 256, 512, 1, False)
run(2, 256, 512, 2, True)
run(2, 256, 1024, 2, True)
run(2, 256, 1024, 1, False)
run(4, 128, 512, 1, True)
