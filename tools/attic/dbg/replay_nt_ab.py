"""Replay-to-replay spread of the gradient norms with the token-contraction kernel on / off."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from mm_unet_amd import mfma_gemm
from mm_unet_amd.loss import DICE_BCE_Loss
from test_modules_gpu import _mmnet
DEV = "cuda:0"
for nt in (True, False, True, False):
    mfma_gemm.NT_ENABLED = nt
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 64, 64, generator=gen).to(DEV)
    t = (torch.rand(2, 1, 64, 64, generator=gen) > 0.88).float().to(DEV)
    torch.manual_seed(0)
    m = _mmnet().train()
    loss_fn = DICE_BCE_Loss()
    for _ in range(2):
        loss_fn(m(x), t).backward(); m.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = loss_fn(m(x), t); loss.backward()
    snaps = []
    for _ in range(4):
        g.replay(); torch.cuda.synchronize()
        snaps.append({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    keys = [k for k in snaps[0] if k.endswith(("mamba.in_proj.weight", "mamba.out_proj.weight", "altho"))]
    n0 = torch.stack([snaps[0][k].norm() for k in keys])
    out = []
    for r in (1, 2, 3):
        nr = torch.stack([snaps[r][k].norm() for k in keys])
        out.append(float(((nr - n0).abs() / (n0 + 1e-12)).median()))
    print("NT", nt, "median rel change per replay", ["%.4f" % v for v in out], flush=True)
