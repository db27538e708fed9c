"""f2 -- the tri-directional Mamba block with its token orders folded into the conv1d and the gate (csrc/tri_fused.hip,
mm-unet_amd/tri_inner.py) against (a) float64 torch compositions of what requirements/mamba_simple.py:212-270 computes,
(b) the REFERENCE's own Mamba(v3) fixture, (c) the three-call route of this package."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _slice_order(t, ns):
    """token i of slice s -> position i*ns + s (mamba_simple.py:245-247)."""
    *lead, L = t.shape
    return t.reshape(*lead, ns, L // ns).transpose(-1, -2).reshape(*lead, L)


def _unslice(t, ns):
    *lead, L = t.shape
    return t.reshape(*lead, L // ns, ns).transpose(-1, -2).reshape(*lead, L)


def _conv_silu(x, w, b):
    """causal depthwise conv1d, width 4, + SiLU (causal_conv1d_interface.py:49-68), any float type."""
    D = x.shape[1]
    y = F.conv1d(x, w.view(D, 1, 4), b, padding=3, groups=D)[..., :x.shape[-1]]
    return F.silu(y)


def _cbl(t):
    """same values, storage [D][B][L]."""
    return t.permute(1, 0, 2).contiguous().permute(1, 0, 2)


@pytest.mark.parametrize("B,D,L,ns,layout", [(2, 6, 4 * 100, 4, "bc"), (3, 5, 16 * 70, 16, "cb"), (2, 4, 64 * 64, 64, "cb"),
                                             (1, 3, 5 * 13, 5, "bc"), (2, 8, 32 * 128, 32, "cb"), (1, 2, 64 * 3, 64, "bc")])
def test_tri_conv_kernels_vs_float64(B, D, L, ns, layout):
    from mm_unet_amd import tri_inner
    gen = torch.Generator().manual_seed(B * 1000 + L + ns)
    xz = torch.randn(B, 2 * D, L, generator=gen)
    if layout == "cb":
        xz = _cbl(xz)
    xz = xz.to(DEV)
    x = xz[:, :D]
    ws = [(0.5 * torch.randn(D, 4, generator=gen)).to(DEV) for _ in range(3)]
    bs = [(0.5 * torch.randn(D, generator=gen)).to(DEV), None, (0.5 * torch.randn(D, generator=gen)).to(DEV)]
    outs = tri_inner.tri_conv_fwd(x, ns, ws, bs)
    # float64 truth with autograd
    xd = x.double().detach().requires_grad_()
    wd = [w.double().requires_grad_() for w in ws]
    bd = [None if b is None else b.double().requires_grad_() for b in bs]
    refs = [_conv_silu(xd, wd[0], bd[0]), _conv_silu(xd.flip(-1), wd[1], bd[1]), _conv_silu(_slice_order(xd, ns), wd[2], bd[2])]
    for k, (o, r) in enumerate(zip(outs, refs)):
        err = (o.double() - r).abs().max().item()
        assert err < 2e-6, f"conv output {k}: {err:.3e}"
    douts = [_cbl(torch.randn(B, D, L, generator=gen)).to(DEV) for _ in range(3)]
    loss = sum((r * g.double()).sum() for r, g in zip(refs, douts))
    loss.backward()
    dxz = torch.full_like(xz, float("nan"))
    dws, dbs = tri_inner.tri_conv_bwd(x, ns, ws, bs, douts, dxz[:, :D])
    torch.cuda.synchronize()
    scale = xd.grad.abs().max().item()
    assert (dxz[:, :D].double() - xd.grad).abs().max().item() < 3e-6 * max(scale, 1.0), "dx"
    assert torch.isnan(dxz[:, D:]).all(), "dx wrote outside the x half"
    for k in range(3):
        e = (dws[k].double() - wd[k].grad).abs().max().item()
        assert e < 2e-5 * max(1.0, wd[k].grad.abs().max().item()), f"dweight {k}: {e:.3e}"
        if bs[k] is None:
            assert dbs[k] is None
        else:
            e = (dbs[k].double() - bd[k].grad).abs().max().item()
            assert e < 2e-5 * max(1.0, bd[k].grad.abs().max().item()), f"dbias {k}: {e:.3e}"


@pytest.mark.parametrize("B,D,L,ns", [(2, 6, 4 * 100, 4), (3, 5, 16 * 70, 16), (2, 4, 64 * 64, 64), (1, 3, 7 * 9, 7)])
def test_tri_gate_kernels_vs_float64(B, D, L, ns):
    from mm_unet_amd import tri_inner
    gen = torch.Generator().manual_seed(L + ns)
    xz = _cbl(torch.randn(B, 2 * D, L, generator=gen)).to(DEV)
    z = xz[:, D:]
    ys = [_cbl(torch.randn(B, D, L, generator=gen)).to(DEV) for _ in range(3)]
    out = tri_inner.tri_gate_fwd(z, ns, ys)
    zd = z.double().detach().requires_grad_()
    yd = [y.double().requires_grad_() for y in ys]
    ref = F.silu(zd) * (yd[0] + yd[1].flip(-1) + _unslice(yd[2], ns))
    assert (out.double() - ref).abs().max().item() < 5e-6
    dout = torch.randn(B, D, L, generator=gen).to(DEV)        # batch-major on purpose: strides are parameters
    (ref * dout.double()).sum().backward()
    dxz = torch.full_like(xz, float("nan"))
    dys = tri_inner.tri_gate_bwd(z, ns, ys, dout, dxz[:, D:])
    torch.cuda.synchronize()
    assert (dxz[:, D:].double() - zd.grad).abs().max().item() < 2e-5
    assert torch.isnan(dxz[:, :D]).all()
    for k in range(3):
        assert (dys[k].double() - yd[k].grad).abs().max().item() < 5e-6, f"dy {k}"


def _load(m, g):
    sd = {k[3:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("sd.")}
    m.load_state_dict(sd, strict=True)
    return m.to(DEV)


def _close(a, b, rtol, atol, what):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, what
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{what}: max abs err {err:.3e} (ref max {b.abs().max():.3e})"


def test_fused_tri_block_vs_reference_fixture():
    """The reference's own Mamba(bimamba_type='v3') output and every gradient (fixture mamba_v3_d64), computed by the
    fused route (no o_1..o_3 wanted)."""
    from mm_unet_amd import tri_inner
    from mm_unet_amd.mamba_simple import Mamba
    g = golden("mamba_v3_d64")
    m = _load(Mamba(int(g["d_model"]), d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=int(g["nslices"])), g)
    m.return_branch_outputs = False
    x = torch.from_numpy(g["x"]).to(DEV).requires_grad_()
    calls = []
    orig = tri_inner.tri_mamba_inner
    tri_inner.tri_mamba_inner = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        out, o1, o2, o3 = m(x)
    finally:
        tri_inner.tri_mamba_inner = orig
    assert calls, "the fused route was not taken"
    assert o1 is None and o2 is None and o3 is None
    _close(out, g["out"], 1e-4, 1e-4, "out")
    out.backward(torch.from_numpy(g["dout"]).to(DEV))
    _close(x.grad, g["dx"], 1e-3, 1e-4, "dx")
    params = dict(m.named_parameters())
    for k in g:
        if k.startswith("grad."):
            _close(params[k[5:]].grad, g[k], 2e-3, 2e-3, k)
    assert {k for k, p in params.items() if p.grad is not None} == {k[5:] for k in g if k.startswith("grad.")}


@pytest.mark.parametrize("B,d_model,L,ns,bcl", [(2, 8, 16 * 24, 16, False), (4, 64, 2048, 32, True), (2, 64, 64 * 64, 64, True)])
def test_fused_tri_block_equals_three_call_route(B, d_model, L, ns, bcl):
    from mm_unet_amd import tri_inner
    from mm_unet_amd.mamba_simple import Mamba
    torch.manual_seed(3)
    m = Mamba(d_model, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=ns).to(DEV)
    m.return_branch_outputs = False
    with torch.no_grad():   # directions that differ from each other more than the default init makes them
        for n, p in m.named_parameters():
            if "A_" in n or n.startswith("D"):
                p.mul_(1.0 + 0.3 * torch.rand_like(p))
    gen = torch.Generator().manual_seed(5)
    x0 = torch.randn(B, d_model, L, generator=gen).to(DEV) if bcl else torch.randn(B, L, d_model, generator=gen).to(DEV)
    dout = torch.randn(x0.shape, generator=gen).to(DEV)
    res = {}
    for fused in (True, False):
        tri_inner.ENABLED = fused
        try:
            for p in m.parameters():
                p.grad = None
            x = x0.clone().requires_grad_()
            out = (m.forward_bcl(x) if bcl else m(x))[0]
            out.backward(dout)
            res[fused] = (out.detach(), x.grad, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            tri_inner.ENABLED = True
    a, b = res[True], res[False]
    _close(a[0], b[0], 2e-4, 2e-5, "out")
    _close(a[1], b[1], 1e-3, 1e-4 * float(b[1].abs().max()), "dx")
    assert set(a[2]) == set(b[2])
    for n in a[2]:
        scale = float(b[2][n].abs().max())
        _close(a[2][n], b[2][n], 2e-3, 2e-4 * max(scale, 1e-3), n)


def test_fused_tri_block_inside_a_deferred_scope():
    """The three conv weight sums join the deferred launch (kind-2 jobs) and land in the parameters' .grad."""
    from mm_unet_amd import deferred
    from mm_unet_amd.mamba_simple import Mamba, precomputed_A
    torch.manual_seed(4)
    m = Mamba(16, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=16).to(DEV)
    m.return_branch_outputs = False
    x = torch.randn(2, 16 * 32, 16, device=DEV)
    dout = torch.randn(2, 16 * 32, 16, device=DEV)
    grads = {}
    for use_scope in (False, True):
        for p in m.parameters():
            p.grad = None
        with precomputed_A(m):
            out = m(x)[0]
            if use_scope:
                scope = deferred.Scope(DEV)
                with scope:
                    out.backward(dout)
                    scope.launch()
                assert scope.n_jobs >= 3
                scope.verify_destinations(list(m.named_parameters()))
            else:
                out.backward(dout)
        torch.cuda.synchronize()
        grads[use_scope] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    for n in grads[False]:
        assert torch.equal(grads[False][n], grads[True][n]), n


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 46, 112), (3, 16, 16), (1, 130, 208)])
def test_stem7_mfma_vs_float64(B, H, W):
    """csrc/stem7_mfma.hip -- Conv2d(3, 64, 7, stride 2, padding 3, no bias): forward and weight gradient against
    float64 F.conv2d (three-part bf16 split = float32-grade: tighter than any float32 convolution needs)."""
    from mm_unet_amd import stem7
    gen = torch.Generator().manual_seed(H * W + B)
    conv = torch.nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False).to(DEV)
    x = torch.randn(B, 3, H, W, generator=gen).to(DEV)
    assert stem7.supported(conv, x)
    out = stem7.stem_conv(conv, x)
    wd = conv.weight.detach().double().requires_grad_()
    ref = F.conv2d(x.double(), wd, None, stride=2, padding=3)
    assert out.shape == ref.shape
    err = float((out.double() - ref).abs().max())
    assert err < 5e-6 * float(ref.abs().max()), err
    dout = torch.randn(ref.shape, generator=gen).to(DEV)
    (ref * dout.double()).sum().backward()
    out.backward(dout)
    e = float((conv.weight.grad.double() - wd.grad).abs().max())
    assert e < 5e-6 * float(wd.grad.abs().max()), e


@pytest.mark.parametrize("B,D,L,ns", [(2, 6, 16 * 70, 16), (2, 4, 64 * 64, 64), (1, 3, 5 * 13, 5)])
def test_tri_kernels_bfloat16_io(B, D, L, ns):
    """bfloat16 activations (float32 arithmetic and weights): each kernel against the float64 composition of ITS bf16
    inputs, within bf16 rounding of the outputs."""
    from mm_unet_amd import tri_inner
    gen = torch.Generator().manual_seed(L + ns)
    bf = torch.bfloat16
    xz = _cbl(torch.randn(B, 2 * D, L, generator=gen)).to(DEV).to(bf)
    x, z = xz[:, :D], xz[:, D:]
    ws = [(0.5 * torch.randn(D, 4, generator=gen)).to(DEV) for _ in range(3)]
    bs = [(0.5 * torch.randn(D, generator=gen)).to(DEV) for _ in range(3)]
    outs = tri_inner.tri_conv_fwd(x, ns, ws, bs)
    xd = x.double().requires_grad_()
    wd = [w.double().requires_grad_() for w in ws]
    bd = [b.double().requires_grad_() for b in bs]
    refs = [_conv_silu(xd, wd[0], bd[0]), _conv_silu(xd.flip(-1), wd[1], bd[1]), _conv_silu(_slice_order(xd, ns), wd[2], bd[2])]
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max().clamp_min(1e-6))   # noqa: E731
    for o, r in zip(outs, refs):
        assert o.dtype == bf and rel(o, r) < 6e-3
    douts = [_cbl(torch.randn(B, D, L, generator=gen)).to(DEV).to(bf) for _ in range(3)]
    sum((r * g.double()).sum() for r, g in zip(refs, douts)).backward()
    dxz = torch.zeros_like(xz)
    dws, dbs = tri_inner.tri_conv_bwd(x, ns, ws, bs, douts, dxz[:, :D])
    assert rel(dxz[:, :D], xd.grad) < 6e-3
    for k in range(3):
        assert dws[k].dtype == torch.float32 and rel(dws[k], wd[k].grad) < 1e-4 and rel(dbs[k], bd[k].grad) < 1e-4
    ys = [_cbl(torch.randn(B, D, L, generator=gen)).to(DEV).to(bf) for _ in range(3)]
    out = tri_inner.tri_gate_fwd(z, ns, ys)
    zd = z.double().requires_grad_()
    yd = [y.double().requires_grad_() for y in ys]
    ref = F.silu(zd) * (yd[0] + yd[1].flip(-1) + _unslice(yd[2], ns))
    assert out.dtype == bf and rel(out, ref) < 6e-3
    dout = torch.randn(B, D, L, generator=gen).to(DEV).to(bf)
    (ref * dout.double()).sum().backward()
    dys = tri_inner.tri_gate_bwd(z, ns, ys, dout, dxz[:, D:])
    assert rel(dxz[:, D:], zd.grad) < 8e-3
    for k in range(3):
        assert rel(dys[k], yd[k].grad) < 6e-3


def test_fused_tri_block_under_bf16_autocast():
    """Mamba(v3).forward_bcl under bf16 autocast: the fused route and the three-call route are both bf16 computations of
    the float32 result; the fused one must not be further from it."""
    from mm_unet_amd import tri_inner
    from mm_unet_amd.mamba_simple import Mamba
    torch.manual_seed(3)
    m = Mamba(64, d_state=16, d_conv=4, expand=2, bimamba_type="v3", nslices=32).to(DEV)
    m.return_branch_outputs = False
    gen = torch.Generator().manual_seed(5)
    x0 = torch.randn(4, 64, 2048, generator=gen).to(DEV)
    dout = torch.randn(x0.shape, generator=gen).to(DEV)

    def run(autocast, fused):
        tri_inner.ENABLED = fused
        try:
            for p in m.parameters():
                p.grad = None
            x = x0.clone().requires_grad_()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                out = m.forward_bcl(x)[0]
            out.float().backward(dout)
            return out.detach().float(), x.grad.float(), {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            tri_inner.ENABLED = True

    calls = []
    orig = tri_inner.tri_mamba_inner
    tri_inner.tri_mamba_inner = lambda *a: (calls.append(a[0].dtype), orig(*a))[1]
    try:
        truth, fused, three = run(False, True), run(True, True), run(True, False)
    finally:
        tri_inner.tri_mamba_inner = orig
    assert torch.bfloat16 in calls, "the fused route was not taken under autocast"
    rms = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12))   # noqa: E731
    assert rms(fused[0], truth[0]) < 1.3 * rms(three[0], truth[0]) + 1e-3, (rms(fused[0], truth[0]), rms(three[0], truth[0]))
    assert rms(fused[1], truth[1]) < 1.3 * rms(three[1], truth[1]) + 1e-3
    assert set(fused[2]) == set(truth[2]) == set(three[2])
    for n in truth[2]:
        ef, et = rms(fused[2][n], truth[2][n]), rms(three[2][n], truth[2][n])
        assert ef < 1.5 * et + 2e-2, (n, ef, et)
