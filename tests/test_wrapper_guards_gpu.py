"""Negative tests for every Python wrapper that hands raw device pointers to a float32 HIP kernel.

Round 1 lost a test process to ``Fatal Python error: Aborted``: the first version of the fused
residual tail passed the bf16 shortcut tensor of an autocast run to a kernel that reads float32 -- twice the
buffer, an out-of-bounds read, a GPU memory fault reported at the next ATen call.  The rule since: a wrapper takes
``data_ptr()`` only of tensors it has itself brought to (or checked to be) contiguous float32 of the expected shape.
Here every wrapper is fed a bf16 tensor, a non-contiguous view and a wrong shape; it must either raise
``RuntimeError`` or return what the float32 / contiguous call returns -- never launch on the foreign buffer."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).to(DEV)


def _same_or_raises(fn, ref, tol=0.0):
    """fn() raises RuntimeError, or returns ref (within tol, after conversion to float32)."""
    try:
        out = fn()
    except RuntimeError:
        return "raised"
    torch.cuda.synchronize()
    outs = out if isinstance(out, (tuple, list)) else (out,)
    refs = ref if isinstance(ref, (tuple, list)) else (ref,)
    for o, r in zip(outs, refs):
        assert o.shape == r.shape
        assert float((o.float() - r.float()).abs().max()) <= tol * max(1.0, float(r.abs().max())), "wrong result"
    return "converted"


def test_norm_fused_guards():
    from mm_unet_amd import norm_fused
    gn, bn = torch.nn.GroupNorm(2, 8).to(DEV), torch.nn.BatchNorm2d(8).to(DEV).eval()
    x = _rnd(2, 8, 6, 10)
    ref = norm_fused.gn_bn_act(x, gn, bn, "relu")
    assert _same_or_raises(lambda: norm_fused.gn_bn_act(x.bfloat16(), gn, bn, "relu"), ref, 2e-2) in ("raised", "converted")
    assert norm_fused.supported(x.bfloat16(), gn, bn)              # bf16 activations are read natively (round 2)
    assert not norm_fused.supported(x.half(), gn, bn)              # what mmunet.run_fused consults first
    with pytest.raises(RuntimeError):
        norm_fused.gn_bn_act(x.half(), gn, bn, "relu")
    xc = x.to(memory_format=torch.channels_last)
    assert _same_or_raises(lambda: norm_fused.gn_bn_act(xc, gn, bn, "relu"), ref, 1e-6) == "converted"
    with pytest.raises(RuntimeError):                               # 6 channels into 8-channel norms
        norm_fused.gn_bn_act(x[:, :6].contiguous(), gn, bn, "relu")
    # the residual tail: bf16 shortcut (what autocast produces) and a wrong-shaped one
    res = _rnd(2, 8, 6, 10, seed=1)
    ref_r = norm_fused.gn_bn_act(x, gn, bn, "relu", residual=res)
    assert _same_or_raises(lambda: norm_fused.gn_bn_act(x, gn, bn, "relu", residual=res.bfloat16()), ref_r, 2e-2) == "converted"
    with pytest.raises(RuntimeError):
        norm_fused.gn_bn_act(x, gn, bn, "relu", residual=res[:, :, :3])
    ref_b = norm_fused.bn_act(x, bn, "relu")
    assert _same_or_raises(lambda: norm_fused.bn_act(x.bfloat16(), bn, "relu"), ref_b, 2e-2) in ("raised", "converted")
    # the incoming gradient may be bf16 / non-contiguous: backward normalises it
    xg = x.clone().requires_grad_()
    out = norm_fused.gn_bn_act(xg, gn, bn, "relu")
    out.backward(torch.ones_like(out).to(memory_format=torch.channels_last))
    g1 = xg.grad.clone()
    xg.grad = None
    norm_fused.gn_bn_act(xg, gn, bn, "relu").backward(torch.ones_like(out))
    assert torch.allclose(g1, xg.grad, atol=1e-6)


def test_conv3x3_small_guards():
    from mm_unet_amd.conv3x3_small import conv3x3_small
    x, w, b = _rnd(2, 16, 9, 12), _rnd(6, 16, 3, 3, seed=1), _rnd(6, seed=2)
    ref = F.conv2d(x, w, b, padding=1)
    assert _same_or_raises(lambda: conv3x3_small(x, w, b), ref, 1e-5) == "converted"
    assert _same_or_raises(lambda: conv3x3_small(x.bfloat16(), w, b), ref, 3e-2) in ("raised", "converted")
    assert _same_or_raises(lambda: conv3x3_small(x.to(memory_format=torch.channels_last), w, b), ref, 1e-5) == "converted"
    assert _same_or_raises(lambda: conv3x3_small(x, w.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2), b), ref, 1e-5) == "converted"
    with pytest.raises(RuntimeError):
        conv3x3_small(x, _rnd(6, 8, 3, 3), b)                       # Cin mismatch
    with pytest.raises(RuntimeError):
        conv3x3_small(x, w, _rnd(5))                                # bias length


def test_morph_sample_and_coords_guards():
    from mm_unet_amd import morph_coords
    from mm_unet_amd.morph_sample import morph_sample
    x, y = _rnd(2, 5, 8, 12), 3.5 + _rnd(2, 3, 8, 12, seed=1)
    ref = morph_sample(x, y)
    assert _same_or_raises(lambda: morph_sample(x.bfloat16(), y), ref, 3e-2) == "converted"
    assert _same_or_raises(lambda: morph_sample(x.to(memory_format=torch.channels_last), y), ref, 1e-6) == "converted"
    with pytest.raises(RuntimeError):
        morph_sample(x, y[:, :, :4])
    with pytest.raises(RuntimeError):
        morph_sample(x, y[:, :2])                                   # even number of taps
    off, w_in = _rnd(2, 6, 8, 12, seed=2), _rnd(12, 3, seed=3)
    ref = morph_coords.zigzag_inproj(off, w_in)
    assert _same_or_raises(lambda: morph_coords.zigzag_inproj(off.bfloat16(), w_in), ref, 3e-2) == "converted"
    assert _same_or_raises(lambda: morph_coords.zigzag_inproj(off.to(memory_format=torch.channels_last), w_in), ref, 1e-6) == "converted"
    with pytest.raises(RuntimeError):
        morph_coords.zigzag_inproj(off, _rnd(12, 4))                # wrong projection shape
    with pytest.raises(RuntimeError):
        morph_coords.zigzag_inproj(off[:, :4], w_in)                # K = 2
    oz, w_out, al = _rnd(2, 6, 96, seed=4), _rnd(3, 6, seed=5), torch.tensor(0.54, device=DEV)
    ref = morph_coords.coords_outproj(off, oz, w_out, al)
    assert _same_or_raises(lambda: morph_coords.coords_outproj(off.bfloat16(), oz.bfloat16(), w_out, al), ref, 5e-2) == "converted"
    with pytest.raises(RuntimeError):
        morph_coords.coords_outproj(off, oz[:, :, :50], w_out, al)  # token count != H*W


def test_resize_and_tri_order_guards():
    from mm_unet_amd.resize import bilinear_resize
    from mm_unet_amd.tri_order import tri_combine, tri_split
    x = _rnd(2, 3, 7, 9)
    ref = F.interpolate(x, size=(14, 18), mode="bilinear", align_corners=True)
    assert _same_or_raises(lambda: bilinear_resize(x, size=(14, 18)), ref, 1e-5) == "converted"
    assert _same_or_raises(lambda: bilinear_resize(x.bfloat16(), size=(14, 18)), ref, 3e-2) == "converted"
    assert _same_or_raises(lambda: bilinear_resize(x.to(memory_format=torch.channels_last), size=(14, 18)), ref, 1e-5) == "converted"
    with pytest.raises(RuntimeError):
        bilinear_resize(x[0], size=(14, 18))
    t = _rnd(2, 4, 64)
    a, f, s = tri_split(t, 4)
    assert torch.equal(f, t.flip(-1))
    assert _same_or_raises(lambda: tri_split(t.bfloat16(), 4), (a, f, s), 3e-2) in ("raised", "converted")
    assert _same_or_raises(lambda: tri_split(t.transpose(1, 2).contiguous().transpose(1, 2), 4), (a, f, s), 0.0) in ("raised", "converted")
    with pytest.raises(RuntimeError):
        tri_split(t, 5)                                             # L not divisible by nslices
    ref = tri_combine(a, f, s, 4)
    assert _same_or_raises(lambda: tri_combine(a, f.bfloat16(), s, 4), ref, 3e-2) in ("raised", "converted")


def test_mfma_wrappers_guards():
    from mm_unet_amd import mfma_gemm
    from mm_unet_amd.conv3x3_mfma import Conv3x3MfmaFn, supported
    x, w = _rnd(1, 16, 8, 16), _rnd(64, 16, 3, 3, seed=1)
    ref = F.conv2d(x, w, None, padding=1)
    assert _same_or_raises(lambda: Conv3x3MfmaFn.apply(x, w, None), ref, 1e-4) == "converted"
    assert not supported(x.bfloat16(), w)
    assert _same_or_raises(lambda: Conv3x3MfmaFn.apply(x.bfloat16(), w, None), ref, 3e-2) in ("raised", "converted")
    assert _same_or_raises(lambda: Conv3x3MfmaFn.apply(x.to(memory_format=torch.channels_last), w, None), ref, 1e-4) in ("raised", "converted")
    with pytest.raises(RuntimeError):
        Conv3x3MfmaFn.apply(x, _rnd(64, 32, 3, 3), None)
    rows, inner, tokens = 64, 32, 1024
    wt, xm, out = _rnd(rows, inner), _rnd(inner, tokens, seed=2), torch.empty(rows, tokens, device=DEV)
    mfma_gemm.gemm_tokens(wt, xm, out, rows, inner, tokens, 1, tokens, 0, tokens, 0)
    torch.cuda.synchronize()
    assert float((out - wt @ xm).abs().max()) < 1e-3
    with pytest.raises(RuntimeError):
        mfma_gemm.gemm_tokens(wt.bfloat16(), xm, out, rows, inner, tokens, 1, tokens, 0, tokens, 0)
    with pytest.raises(RuntimeError):
        mfma_gemm.gemm_tokens(wt, xm.bfloat16(), out, rows, inner, tokens, 1, tokens, 0, tokens, 0)
    with pytest.raises(RuntimeError):
        mfma_gemm.gemm_tokens(wt[:, :16], xm, out, rows, inner, tokens, 1, tokens, 0, tokens, 0)
    assert not mfma_gemm.supported(rows, inner, tokens, wt.bfloat16(), xm, out)


def test_conv1d_and_scan_extension_guards():
    from mm_unet_amd import causal_conv1d_hip as cc, selective_scan_hip as ss
    x, w, b = _rnd(2, 8, 64), _rnd(8, 4, seed=1), _rnd(8, seed=2)
    ref = cc.causal_conv1d_fwd(x, w, b, True)
    with pytest.raises(RuntimeError):
        cc.causal_conv1d_fwd(x.half(), w.half(), b.half(), True)    # no float16 I/O
    with pytest.raises(RuntimeError):
        cc.causal_conv1d_fwd(x, w[:6], b, True)                     # weight rows != channels
    with pytest.raises(RuntimeError):
        cc.causal_conv1d_fwd(x, _rnd(8, 5), b, True)                # width 5
    with pytest.raises(RuntimeError):
        cc.causal_conv1d_bwd(x, w, b, _rnd(2, 8, 32), None, True)   # dout length
    assert _same_or_raises(lambda: cc.causal_conv1d_fwd(x.transpose(0, 1).contiguous().transpose(0, 1), w, b, True), ref, 1e-6) == "converted"
    u, delta = _rnd(1, 4, 64), 0.5 * torch.rand(1, 4, 64, device=DEV)
    A, B, C = -torch.rand(4, 16, device=DEV), _rnd(1, 1, 16, 64, seed=3), _rnd(1, 1, 16, 64, seed=4)
    with pytest.raises(RuntimeError):
        ss.fwd(u, delta.bfloat16(), A, B, C, None, None, None, False)   # mixed I/O dtypes
    with pytest.raises(RuntimeError):
        ss.fwd(u, delta, A.bfloat16(), B, C, None, None, None, False)   # A must be float32
    with pytest.raises(RuntimeError):
        ss.fwd(u, delta, A, B[..., :32], C, None, None, None, False)    # B length
    with pytest.raises(RuntimeError):
        ss.fwd(u, delta, A, B, C, torch.zeros(3, device=DEV), None, None, False)  # D length
    out, xs = ss.fwd(u, delta, A, B, C, None, None, None, False)
    with pytest.raises(RuntimeError):
        ss.bwd(u, delta, A, B, C, None, None, None, _rnd(1, 4, 32), xs, out, None, False, False)  # dout length


def test_bf16_activations_are_read_and_written_natively():
    """Round 2: bf16 as an I/O type of the fused MorphMamba chain.  The kernels read bf16
    activations as they are and compute in float32, so on bf16-representable inputs they must agree with the
    float32 call to rounding of the OUTPUT only; gradients come back in the input's type."""
    from mm_unet_amd import norm_fused
    from mm_unet_amd.conv3x3_small import conv3x3_small
    from mm_unet_amd.morph_sample import morph_sample

    def pair(*shape, seed=0):
        a = _rnd(*shape, seed=seed).bfloat16()
        return a, a.float()

    # conv3x3_small: bf16 input, float32 weights / output
    xb, xf = pair(2, 16, 8, 12)
    w, b = _rnd(6, 16, 3, 3, seed=1), _rnd(6, seed=2)
    xb.requires_grad_(); xf.requires_grad_()
    wb, wf = w.clone().requires_grad_(), w.clone().requires_grad_()
    ob, of = conv3x3_small(xb, wb, b), conv3x3_small(xf, wf, b)
    assert ob.dtype == torch.float32 and torch.allclose(ob, of, atol=1e-5)
    g = _rnd(*of.shape, seed=3)
    ob.backward(g); of.backward(g)
    assert xb.grad.dtype == torch.bfloat16
    assert torch.allclose(xb.grad.float(), xf.grad, atol=2e-2 * float(xf.grad.abs().max()))
    assert torch.allclose(wb.grad, wf.grad, atol=1e-4 * float(wf.grad.abs().max()))
    # morph_sample: bf16 input, float32 coordinates / samples
    xb, xf = pair(2, 5, 8, 12, seed=4)
    y = 3.5 + _rnd(2, 3, 8, 12, seed=5)
    xb.requires_grad_(); xf.requires_grad_()
    yb, yf = y.clone().requires_grad_(), y.clone().requires_grad_()
    for tl in (False, True):
        sb, sf = morph_sample(xb, yb, tokens_last=tl), morph_sample(xf, yf, tokens_last=tl)
        assert sb.dtype == torch.float32 and torch.allclose(sb, sf, atol=1e-6)
        g = _rnd(*sf.shape, seed=6)
        for t_ in (xb, xf, yb, yf):
            t_.grad = None
        sb.backward(g); sf.backward(g)
        assert xb.grad.dtype == torch.bfloat16
        assert torch.allclose(xb.grad.float(), xf.grad, atol=2e-2 * float(xf.grad.abs().max()))
        assert torch.allclose(yb.grad, yf.grad, atol=1e-5 * float(yf.grad.abs().max()) + 1e-6)
    # bilinear_resize: bf16 in -> bf16 out, interpolated in float32
    from mm_unet_amd.resize import bilinear_resize
    xb, xf = pair(2, 3, 9, 14, seed=10)
    xb.requires_grad_(); xf.requires_grad_()
    rb, rf = bilinear_resize(xb, size=(18, 28)), bilinear_resize(xf, size=(18, 28))
    assert rb.dtype == torch.bfloat16 and torch.allclose(rb.float(), rf, atol=1e-2 * float(rf.abs().max()))
    g = _rnd(*rf.shape, seed=11).bfloat16()
    rb.backward(g); rf.backward(g.float())
    assert xb.grad.dtype == torch.bfloat16
    assert torch.allclose(xb.grad.float(), xf.grad, atol=1e-2 * float(xf.grad.abs().max()))
    # gn_bn_act: float32 in -> bf16 out under autocast; bf16 in -> bf16 out; residual in either type
    gn, bn = torch.nn.GroupNorm(2, 8).to(DEV), torch.nn.BatchNorm2d(8).to(DEV).train()
    x32 = _rnd(2, 8, 6, 12, seed=7).requires_grad_()
    res = _rnd(2, 8, 6, 12, seed=8)
    ref = norm_fused.gn_bn_act(x32, gn, bn, "relu", residual=res)
    gref = _rnd(*ref.shape, seed=9)
    ref.backward(gref)
    gx_ref, x32.grad = x32.grad.clone(), None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = norm_fused.gn_bn_act(x32, gn, bn, "relu", residual=res.bfloat16())
        off = norm_fused.gn_bn_act(x32, gn, None, "tanh", out_dtype=torch.float32)
    assert out.dtype == torch.bfloat16 and off.dtype == torch.float32
    assert torch.allclose(out.float(), ref, atol=3e-2)
    out.backward(gref.bfloat16())
    assert x32.grad.dtype == torch.float32
    # (the bf16-rounded residual flips the ReLU mask of the few elements whose pre-activation is within rounding of 0)
    off_tol = ((x32.grad - gx_ref).abs() > 3e-2 * float(gx_ref.abs().max())).float().mean()
    assert float(off_tol) < 0.01, float(off_tol)
    xb16 = x32.detach().bfloat16().requires_grad_()
    ob16 = norm_fused.gn_bn_act(xb16, gn, bn, "relu")
    assert ob16.dtype == torch.bfloat16
    ref16 = norm_fused.gn_bn_act(xb16.detach().float(), gn, bn, "relu")
    assert torch.allclose(ob16.float(), ref16, atol=2e-2)
    ob16.backward(gref.bfloat16())
    assert xb16.grad.dtype == torch.bfloat16 and torch.isfinite(xb16.grad.float()).all()


def test_pointwise_maxpool_gemm_nt_sum_parts_guards():
    """the round-2 wrappers (CBAM statistics, gated products, max-pool backward, token-contraction GEMM, the
    state groups' partial sums) under the same rule: bf16 / wrong-rank / wrong-gate inputs raise (or give the float32
    result), never reach a float32 kernel as they are."""
    from mm_unet_amd import maxpool, mfma_gemm, pointwise
    from mm_unet_amd import selective_scan_hip as ss
    x = _rnd(2, 8, 8, 8)
    mean, mx = pointwise.pixel_mean_max(x)
    torch.cuda.synchronize()
    assert float((mean.flatten() - x.mean((2, 3)).flatten()).abs().max()) < 1e-5
    assert float((mx.flatten() - x.amax((2, 3)).flatten()).abs().max()) == 0.0
    for bad in (x.bfloat16(), x[0], _rnd(2, 8, 3, 5)):
        with pytest.raises(RuntimeError):
            pointwise.pixel_mean_max(bad)
        with pytest.raises(RuntimeError):
            pointwise.channel_max_mean(bad)
    gate_c, gate_s = _rnd(2, 8, 1, 1, seed=1), _rnd(2, 1, 8, 8, seed=2)
    out = pointwise.GatedMulFn.apply(x, gate_c, 0)
    torch.cuda.synchronize()
    assert float((out - x * gate_c).abs().max()) < 1e-6
    for args in ((x.bfloat16(), gate_c, 0), (x, gate_c.bfloat16(), 0), (x, gate_s, 0), (x, gate_c, 1),
                 (x, _rnd(2, 4, 1, 1), 0), (x[0], gate_c, 0)):
        with pytest.raises(RuntimeError):
            pointwise.GatedMulFn.apply(*args)
    assert pointwise._gate_mode(x.bfloat16(), gate_c) is None and pointwise._gate_mode(x, _rnd(2, 4, 1, 1)) is None
    with pytest.raises(RuntimeError):      # bfloat16 maps: W % 8 == 0 and even H only (the vector kernels' bf16 forms)
        maxpool.max_pool3s2(x.bfloat16()[..., :6].contiguous())
    with pytest.raises(RuntimeError):
        maxpool.max_pool3s2(x.bfloat16()[:, :, :7].contiguous())
    assert torch.equal(maxpool.max_pool3s2(x.bfloat16()), torch.nn.functional.max_pool2d(x.bfloat16(), 3, 2, 1))
    with pytest.raises(RuntimeError):
        maxpool.max_pool3s2(x[0])
    a, b = _rnd(8, 256), _rnd(4, 256, seed=3)
    c = mfma_gemm.gemm_nt(a, b, 8, 4, 1, 256, 256, 0, 256, 0)
    torch.cuda.synchronize()
    assert float((c - a @ b.t()).abs().max()) < 1e-3
    with pytest.raises(RuntimeError):
        mfma_gemm.gemm_nt(a.bfloat16(), b, 8, 4, 1, 256, 256, 0, 256, 0)
    with pytest.raises(RuntimeError):
        mfma_gemm.gemm_nt(a, b, 8, 4, 1, 250, 256, 0, 256, 0)
    assert not mfma_gemm.nt_supported(a.bfloat16(), b, 256)
    # partial sums: a bf16 part or parts with different strides take the tensor-op route and still give the sum
    p0, p1 = _rnd(2, 4, 64), _rnd(2, 4, 64, seed=4)
    ref = p0 + p1
    for parts in ([p0, p1], [p0.bfloat16(), p1], [p0, p1.permute(1, 0, 2).contiguous().permute(1, 0, 2)]):
        got = ss._sum_parts([t.clone() for t in parts], torch.float32)
        torch.cuda.synchronize()
        assert float((got.float() - ref).abs().max()) < 5e-2 if parts[0].dtype != torch.float32 else \
            float((got - ref).abs().max()) < 1e-6
