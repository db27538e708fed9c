"""x_proj / dt_proj and their input gradients at the largest RCG block (128 channels, 8 x 65,536 tokens): the build's
kernels (gemm_tokens with a padded 36-row image, dt_proj.hip) and the library GEMMs, N times each (for rocprofv3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd import mfma_gemm as mg
DEV = "cuda:0"
D, R, N, T = 128, 4, 16, int(os.environ.get("PROJ_T", 8 * 65536))
it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator(device=DEV).manual_seed(0)
conv = torch.randn(D, T, device=DEV, generator=g)
Wx = torch.randn(R + 2 * N, D, device=DEV, generator=g) / D ** 0.5
Wdt = torch.randn(D, R, device=DEV, generator=g)
xdbl = torch.empty(R + 2 * N, T, device=DEV)
delta = torch.empty(D, T, device=DEV)
dxdbl = torch.randn(R + 2 * N, T, device=DEV, generator=g)
dconv = torch.zeros(D, T, device=DEV)
for _ in range(it):
    mg.gemm_tokens(Wx, conv, xdbl, R + 2 * N, D, T, 1, T, 0, T, 0)
    mg.x_proj(Wx, conv, xdbl)
    mg.x_proj_input_grad_add(Wx, dxdbl, dconv)
    mg.dt_proj(Wdt, xdbl[:R], delta)
    mg.dt_proj_input_grad(Wdt, delta, dxdbl[:R])
    mg.gemm_tokens(Wx, dxdbl, dconv, D, R + 2 * N, T, 1, T, 0, T, 0, transposed_weight=True, accumulate=True)
    torch.matmul(Wx, conv, out=xdbl)
    torch.matmul(Wdt, xdbl[:R], out=delta)
    torch.matmul(Wdt.t(), delta, out=dxdbl[:R])
    dconv.addmm_(Wx.t(), dxdbl)
torch.cuda.synchronize()
