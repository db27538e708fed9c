"""Generate tests/golden/*.npz from the REFERENCE itself (build container only).

Run:  python tools/make_golden.py [kernels|modules|all]

The reference's pure-torch oracles (selective_scan_ref, causal_conv1d_ref) and --
for the module-level fixtures -- its nn.Modules are imported from /root/reference
through tools/ref_import.py and evaluated on CPU in fp32.  Inputs follow the
reference's own test distributions (tests/ops/test_selective_scan.py:53-88,
tests/test_causal_conv1d.py:36-48): seed 0, A = -0.5*rand, delta = 0.5*rand,
delta_bias = 0.5*rand, everything else randn.  Inputs AND outputs are stored, so
the fixtures do not depend on RNG reproducibility.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

import ref_import  # noqa: E402


def _np(t):
    return None if t is None else t.detach().cpu().numpy().astype(np.float32)


def save(name, **arrs):
    arrs = {k: v for k, v in arrs.items() if v is not None}
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz  {os.path.getsize(path)/1024:.0f} KB")


# (batch, dim, seqlen, dstate, groups, has_z, has_D, has_bias, softplus)
SCAN_CASES = [
    ("reftest_L128_g1", 2, 4, 128, 8, 1, True, True, True, True),
    ("reftest_L512_g2", 2, 4, 512, 8, 2, True, True, True, True),
    ("mmconv_D6_L256", 2, 6, 256, 16, 1, True, True, True, True),
    ("ragged_D4_L2085", 1, 4, 2048 + 37, 16, 1, True, True, True, True),
    ("rcg_D16_L1024", 1, 16, 1024, 16, 1, True, True, True, True),
    ("c5_D8_L512_N64", 1, 8, 512, 64, 1, True, True, True, True),
    ("plain_noz_noD", 2, 4, 300, 8, 1, False, False, False, False),
    ("tiny_L5", 1, 2, 5, 16, 1, True, True, True, True),
]


def make_scan(selective_scan_ref):
    for (name, b, d, l, n, g, has_z, has_D, has_bias, sp) in SCAN_CASES:
        torch.manual_seed(0)
        A = (-0.5 * torch.rand(d, n)).requires_grad_()
        bshape = (b, n, l) if g == 1 else (b, g, n, l)
        B = torch.randn(*bshape, requires_grad=True)
        C = torch.randn(*bshape, requires_grad=True)
        D = torch.randn(d, requires_grad=True) if has_D else None
        z = torch.randn(b, d, l, requires_grad=True) if has_z else None
        bias = (0.5 * torch.rand(d)).requires_grad_() if has_bias else None
        u = torch.randn(b, d, l, requires_grad=True)
        delta = (0.5 * torch.rand(b, d, l)).requires_grad_()
        out, last = selective_scan_ref(u, delta, A, B, C, D, z=z, delta_bias=bias, delta_softplus=sp,
                                       return_last_state=True)
        gout = torch.randn_like(out)
        out.backward(gout)
        save("scan_" + name, u=_np(u), delta=_np(delta), A=_np(A), B=_np(B), C=_np(C), D=_np(D), z=_np(z),
             delta_bias=_np(bias), softplus=np.array(int(sp)), out=_np(out), last_state=_np(last),
             dout=_np(gout), du=_np(u.grad), ddelta=_np(delta.grad), dA=_np(A.grad), dB=_np(B.grad),
             dC=_np(C.grad), dD=_np(D.grad) if has_D else None, dz=_np(z.grad) if has_z else None,
             ddelta_bias=_np(bias.grad) if has_bias else None)


# (batch, dim, seqlen, width, silu, has_bias)
CONV_CASES = [
    ("w4_silu_L8", 2, 8, 8, 4, True, True),
    ("w4_silu_L151", 2, 8, 151, 4, True, True),
    ("w4_silu_L1024", 2, 6, 1024, 4, True, True),
    ("w3_nosilu_L372", 2, 8, 372, 3, False, True),
    ("w2_silu_nobias_L64", 2, 8, 64, 2, True, False),
    ("w4_silu_L2", 1, 4, 2, 4, True, True),
]


def make_conv(causal_conv1d_ref):
    for (name, b, d, l, w, silu, has_bias) in CONV_CASES:
        torch.manual_seed(0)
        x = torch.randn(b, d, l, requires_grad=True)
        weight = torch.randn(d, w, requires_grad=True)
        bias = torch.randn(d, requires_grad=True) if has_bias else None
        out = causal_conv1d_ref(x, weight, bias, activation="silu" if silu else None)
        g = torch.randn_like(out)
        out.backward(g)
        save("conv1d_" + name, x=_np(x), weight=_np(weight), bias=_np(bias), silu=np.array(int(silu)),
             out=_np(out), dout=_np(g), dx=_np(x.grad), dweight=_np(weight.grad),
             dbias=_np(bias.grad) if has_bias else None)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if what in ("kernels", "all"):
        selective_scan_ref, causal_conv1d_ref = ref_import.load_leaf_refs()
        print("scan fixtures (reference selective_scan_ref)")
        make_scan(selective_scan_ref)
        print("conv1d fixtures (reference causal_conv1d_ref)")
        make_conv(causal_conv1d_ref)
    if what in ("modules", "all"):
        import make_golden_modules
        make_golden_modules.main()
    if what == "mmconv":                # python tools/make_golden.py mmconv <fixture name> ...: single MMConv fixtures
        import make_golden_modules
        make_golden_modules.make_mmconv(ref_import.load_reference_model(), only=set(sys.argv[2:]))
    if what in ("mmnet128", "all"):     # the better-conditioned train-mode fixture of round 3 (~30 min of CPU here)
        import make_golden_modules
        R = ref_import.load_reference_model()
        make_golden_modules.make_mmnet_train128(R)


if __name__ == "__main__":
    main()
