"""Runs ONLY the selective-scan forward at the headline shape, N launches: the inference form (no `out`) by default,
PROF_FWD_OUT=1: with the un-gated `out` written too (what mamba_inner's training forward asks for since round 4)."""
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
u = torch.randn(b, d, l, device=DEV, generator=g)
z = torch.randn(b, d, l, device=DEV, generator=g)
delta = 0.5 * torch.rand(b, d, l, device=DEV, generator=g)
torch.cuda.synchronize()
for _ in range(it):
    ss.fwd(u, delta, A, B, C, D, z, bias, True, want_out=os.environ.get("PROF_FWD_OUT") == "1")
torch.cuda.synchronize()
