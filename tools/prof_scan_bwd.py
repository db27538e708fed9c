"""Runs the selective-scan backward (model layout: tokens-last [D][B][L] views) N times at the headline shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd import selective_scan_hip as ss
DEV = "cuda:0"
b, d, l, n = 8, 128, 65536, 16
it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator(device=DEV).manual_seed(0)
A = -0.5 * torch.rand(d, n, device=DEV, generator=g)
B = torch.randn(b, 1, n, l, device=DEV, generator=g)
C = torch.randn(b, 1, n, l, device=DEV, generator=g)
D = torch.randn(d, device=DEV, generator=g)
bias = 0.5 * torch.rand(d, device=DEV, generator=g)
mk = lambda: torch.randn(d, b, l, device=DEV, generator=g).permute(1, 0, 2)
u, z, dout = mk(), mk(), mk()
delta = (0.5 * torch.rand(d, b, l, device=DEV, generator=g)).permute(1, 0, 2)
res = ss.fwd(u, delta, A, B, C, D, z, bias, True)
out = res[0] if os.environ.get("PROF_BWD_OUT", "1") != "0" else None   # the saved y, as mamba_inner hands it over
for _ in range(it):
    ss.bwd(u, delta, A, B, C, D, z, bias, dout, res[1], out, None, True, False)
torch.cuda.synchronize()
