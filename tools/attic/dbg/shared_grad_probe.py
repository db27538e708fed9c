"""Which consumer of a conv3x3_small.SharedGrad disagrees with autograd's sum at a given shape (debug aid)."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mm_unet_amd  # noqa
from mm_unet_amd import conv3x3_small, maxpool, pointwise
from mm_unet_amd.resize import bilinear_resize
DEV = "cuda"
for shape in [(1, 3, 18, 20), (2, 8, 32, 32)]:
    B, C, H, W = shape
    gen = torch.Generator(device=DEV).manual_seed(31)
    x = torch.randn(*shape, device=DEV, generator=gen)
    w = 0.2 * torch.randn(1, C, 3, 3, device=DEV, generator=gen)
    gate = torch.rand(B, C, 1, 1, device=DEV, generator=gen)

    def run(shared, order):
        xi = x.clone().requires_grad_()
        slot = conv3x3_small.SharedGrad() if shared else None
        heads = {
            "r2": lambda: bilinear_resize(xi, size=(H // 2, W // 2), slot=slot),
            "r4": lambda: bilinear_resize(xi, size=(max(H // 4, 1), max(W // 4, 1)), slot=slot),
            "up": lambda: bilinear_resize(xi, size=(H + 3, W + 5), slot=slot),
            "conv": lambda: conv3x3_small.conv3x3_small(xi, w, None, slot),
            "pool": lambda: maxpool.max_pool3s2(xi, slot),
            "join": lambda: conv3x3_small.shared_input(xi, slot) * 0.5,
            "stats": lambda: sum(pointwise.pixel_mean_max(xi, slot)),
            "gate": lambda: pointwise.gated_mul(xi, gate, slot),
        }
        gg = torch.Generator(device=DEV).manual_seed(5)
        total = 0
        outs = {k: heads[k]() for k in order}
        for k in sorted(outs):
            o = outs[k]
            total = total + (o * torch.randn(o.shape, device=DEV, generator=gg)).sum()
        total.backward()
        return xi.grad
    for k in ["r2", "r4", "up", "conv", "pool", "stats", "gate"]:
        for order in ([k, "join"], ["join", k]):
            a, b = run(True, order), run(False, order)
            print(shape, order, float((a - b).abs().max()), float(b.abs().max()))
