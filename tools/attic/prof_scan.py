"""Runs the selective-scan fwd/bwd a few times at the headline shape (for rocprofv3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd import selective_scan_hip as ss
DEV = "cuda:0"
b, d, l, n = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (8, 128, 65536, 16)))
it = int(sys.argv[5]) if len(sys.argv) > 5 else 3
g = torch.Generator(device=DEV).manual_seed(0)
A = -0.5 * torch.rand(d, n, device=DEV, generator=g)
B = torch.randn(b, 1, n, l, device=DEV, generator=g)
C = torch.randn(b, 1, n, l, device=DEV, generator=g)
D = torch.randn(d, device=DEV, generator=g)
bias = 0.5 * torch.rand(d, device=DEV, generator=g)
mk = lambda: torch.randn(d, b, l, device=DEV, generator=g).permute(1, 0, 2)
u, z, dout = mk(), mk(), mk()
delta = (0.5 * torch.rand(d, b, l, device=DEV, generator=g)).permute(1, 0, 2)
for _ in range(it):
    res = ss.fwd(u, delta, A, B, C, D, z, bias, True)
for _ in range(it):
    ss.bwd(u, delta, A, B, C, D, z, bias, dout, res[1], None, None, True, False)
torch.cuda.synchronize()
