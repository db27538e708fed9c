"""Debug: dump fwd/bwd scan outputs for a seeded case to an .npz (compare two library builds)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd import selective_scan_hip as ss
out, b, d, l, n = sys.argv[1], *map(int, sys.argv[2:6])
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
A = -torch.rand(d, n, device="cuda", generator=g)
u, z, dout = r(b, d, l), r(b, d, l), r(b, d, l)
delta = torch.rand(b, d, l, device="cuda", generator=g)
B, C = r(b, 1, n, l), r(b, 1, n, l)
D, bias = r(d), torch.rand(d, device="cuda", generator=g)
res = ss.fwd(u, delta, A, B, C, D, z, bias, True)
_ws = []
_orig_empty = torch.empty
def _rec_empty(*a, **k):
    t = _orig_empty(*a, **k)
    if len(a) == 1 and isinstance(a[0], int):
        t.zero_()
        _ws.append(t)
    return t
ss.torch.empty = _rec_empty
res2 = ss.bwd(u, delta, A, B, C, D, z, bias, dout, res[1], res[0], None, True, True)
res3 = ss.bwd(u, delta, A, B, C, D, z, bias, dout, None, res[0], None, True, True)
torch.cuda.synchronize()
names = ["out", "x", "out_z"] + ["x_" + k for k in "du ddelta dA dB dC dD dbias dz".split()] + ["n_" + k for k in "du ddelta dA dB dC dD dbias dz".split()]
vals = list(res[:3]) + list(res2[:8]) + list(res3[:8])
nc = res[1].shape[2]
names += ["gx_x", "gx_n"]
vals += [_ws[0][: b * d * nc * 2 * n].view(b, d, nc, 2 * n), _ws[1][: b * d * nc * 2 * n].view(b, d, nc, 2 * n)]
np.savez(out, **{k: v.float().cpu().numpy() for k, v in zip(names, vals)})
