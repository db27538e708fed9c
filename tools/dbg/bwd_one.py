"""One scan backward shape, a few calls: for rocprofv3 --kernel-trace --stats (per-kernel split of the backward).
usage: python tools/dbg/bwd_one.py [B D L] [iters]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_unet_amd import selective_scan_hip as ss  # noqa: E402

DEV = "cuda:0"
b, d, l = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 128, 65536)
iters = int(sys.argv[4]) if len(sys.argv) >= 5 else 5
n = 16
g = torch.Generator(device=DEV).manual_seed(0)
A = -0.5 * torch.rand(d, n, device=DEV, generator=g)
B = torch.randn(b, 1, n, l, device=DEV, generator=g)
C = torch.randn(b, 1, n, l, device=DEV, generator=g)
D = torch.randn(d, device=DEV, generator=g)
bias = 0.5 * torch.rand(d, device=DEV, generator=g)
mk = lambda: torch.randn(d, b, l, device=DEV, generator=g).permute(1, 0, 2)  # noqa: E731
u, z, dout = mk(), mk(), mk()
delta = (0.5 * torch.rand(d, b, l, device=DEV, generator=g)).permute(1, 0, 2)
res = ss.fwd(u, delta, A, B, C, D, z, bias, True)
for _ in range(iters):
    ss.bwd(u, delta, A, B, C, D, z, bias, dout, res[1], None, None, True, False)
torch.cuda.synchronize()
print("done")
