"""GPU parity tests for the hand-written kernels, called through the extension-shaped host
modules (which call the C-ABI): HIP result vs (a) the committed golden vectors produced by the
reference's own *_ref functions, (b) the CPU oracle on seeded inputs, (c) size-independent
properties at BASELINE.json's full sizes.

Tolerances are the reference's fp32 ones (tests/ops/test_selective_scan.py:45-51,137-149;
tests/test_causal_conv1d.py:31-34); forward outputs are additionally held to the north-star
bound of 1e-3 absolute.
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu

SCAN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "scan_*.npz")))
CONV = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "conv1d_*.npz")))
RTOL, ATOL = 6e-4, 2e-3
RTOLW, ATOLW = 1e-3, 2e-3
DEV = "cuda:0"


def _g(g, k):
    return torch.from_numpy(g[k]).to(DEV) if k in g else None


def close(a, b, rtol, atol, what):
    a = torch.as_tensor(a).detach().float().cpu()
    b = torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{what}: max abs err {err:.3e}"
    return err


def _bc4(t):
    return t if t.dim() == 4 else t.unsqueeze(1)


# --------------------------------------------------------------------------- wave primitives
@pytest.mark.parametrize("reverse", [0, 1])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_wave_affine_scan(reverse, variant):
    import ctypes
    from mm_unet_amd import _lib
    nw = 7
    gen = torch.Generator().manual_seed(3)
    P = (0.5 + 0.5 * torch.rand(nw * 64, generator=gen)).to(DEV)
    S = torch.randn(nw * 64, generator=gen).to(DEV)
    oP, oS = torch.empty_like(P), torch.empty_like(S)
    _lib.check(_lib.lib().mmu_debug_wave_scan(P.data_ptr(), S.data_ptr(), oP.data_ptr(), oS.data_ptr(), nw,
                                              reverse, variant, _lib.stream_of(P)))
    torch.cuda.synchronize()
    p, s = P.cpu().double().view(nw, 64), S.cpu().double().view(nw, 64)
    if variant == 4:  # first scan of the interleaved pair runs on (0.5*P, -S)
        p, s = 0.5 * p, -s
    if reverse:
        p, s = p.flip(1), s.flip(1)
    ep, es = torch.empty_like(p), torch.empty_like(s)
    cp, cs = torch.ones(nw, dtype=torch.double), torch.zeros(nw, dtype=torch.double)
    for i in range(64):
        cs = p[:, i] * cs + s[:, i]
        cp = cp * p[:, i]
        ep[:, i], es[:, i] = cp, cs
    if reverse:
        ep, es = ep.flip(1), es.flip(1)
    close(oP.view(nw, 64), ep.float(), 1e-5, 1e-6, "P")
    close(oS.view(nw, 64), es.float(), 1e-5, 1e-5, "S")


# --------------------------------------------------------------------------- selective scan
def _run_fwd(g):
    from mm_unet_amd import selective_scan_hip as ss
    u, delta, A, B, C = (_g(g, k) for k in ("u", "delta", "A", "B", "C"))
    res = ss.fwd(u, delta, A, _bc4(B), _bc4(C), _g(g, "D"), _g(g, "z"), _g(g, "delta_bias"), bool(g["softplus"]))
    torch.cuda.synchronize()
    return res


@pytest.mark.parametrize("name", SCAN)
def test_scan_fwd_golden(name):
    g = golden(name)
    res = _run_fwd(g)
    out = res[2] if "z" in g else res[0]
    err = close(out, g["out"], RTOL, ATOL, "out")
    assert err <= 1e-3 * max(1.0, float(np.abs(g["out"]).max())), f"north-star 1e-3 bound exceeded: {err}"
    x = res[1]
    close(x[:, :, -1, 1::2], g["last_state"], RTOL, ATOL, "last_state")


@pytest.mark.parametrize("name", SCAN)
@pytest.mark.parametrize("pass_x", [True, False])
def test_scan_bwd_golden(name, pass_x):
    from mm_unet_amd import selective_scan_hip as ss
    g = golden(name)
    u, delta, A, B, C = (_g(g, k) for k in ("u", "delta", "A", "B", "C"))
    D, z, bias, dout = (_g(g, k) for k in ("D", "z", "delta_bias", "dout"))
    sp = bool(g["softplus"])
    res = ss.fwd(u, delta, A, _bc4(B), _bc4(C), D, z, bias, sp)
    r = ss.bwd(u, delta, A, _bc4(B), _bc4(C), D, z, bias, dout, res[1] if pass_x else None, res[0], None, sp,
               z is not None)
    torch.cuda.synchronize()
    du, ddelta, dA, dB, dC, dD, dbias = r[:7]
    close(du, g["du"], RTOL * 2, ATOL * 2, "du")
    close(ddelta, g["ddelta"], RTOL * 5, ATOL * 10, "ddelta")
    close(dA, g["dA"], RTOLW, ATOLW * 5, "dA")
    close(dB.reshape(g["dB"].shape), g["dB"], RTOL, ATOL, "dB")
    close(dC.reshape(g["dC"].shape), g["dC"], RTOL, ATOL, "dC")
    if "dD" in g:
        close(dD, g["dD"], RTOLW, ATOLW, "dD")
    if "ddelta_bias" in g:
        close(dbias, g["ddelta_bias"], RTOLW, ATOLW, "ddelta_bias")
    if "dz" in g:
        close(r[7], g["dz"], RTOLW, ATOLW, "dz")
        close(r[8], g["out"], RTOL, ATOL, "recomputed out_z")


def _rand_case(b, d, l, n, seed, device=DEV):
    gen = torch.Generator().manual_seed(seed)
    A = -0.5 * torch.rand(d, n, generator=gen)
    B = torch.randn(b, 1, n, l, generator=gen)
    C = torch.randn(b, 1, n, l, generator=gen)
    D = torch.randn(d, generator=gen)
    z = torch.randn(b, d, l, generator=gen)
    bias = 0.5 * torch.rand(d, generator=gen)
    u = torch.randn(b, d, l, generator=gen)
    delta = 0.5 * torch.rand(b, d, l, generator=gen)
    dout = torch.randn(b, d, l, generator=gen)
    return dict(u=u, delta=delta, A=A, B=B, C=C, D=D, z=z, delta_bias=bias, dout=dout)


@pytest.mark.parametrize("shape", [(2, 6, 1024, 16), (2, 6, 4099, 16), (1, 128, 2048, 16), (2, 2, 256, 16),
                                   (1, 8, 777, 64), (1, 4, 300, 128), (3, 5, 130, 8),
                                   # full 512-token tiles with an odd dstate (zero partner state of the packed
                                   # pairs), and >= 64 chunks (parallel chunk-carry kernel) for dstate 8 and 16
                                   (1, 3, 1024, 5), (1, 4, 8192, 8), (1, 3, 16384, 16)])
def test_scan_fwd_bwd_vs_oracle_mamba_layout(shape):
    """Seeded inputs in the layout mamba_inner hands over: u, delta, z are physically [D][B][L]
    (strides (L, B*L, 1)), SURVEY.md section 8a-5."""
    import oracle
    from mm_unet_amd import selective_scan_hip as ss
    b, d, l, n = shape
    c = _rand_case(b, d, l, n, seed=11)

    def dbl(t):  # [D][B][L] physical, viewed as (B, D, L)
        return t.permute(1, 0, 2).contiguous().to(DEV).permute(1, 0, 2)

    u, delta, z, dout = dbl(c["u"]), dbl(c["delta"]), dbl(c["z"]), dbl(c["dout"])
    assert b == 1 or u.stride() == (l, b * l, 1)
    A, B, C, D, bias = (c[k].to(DEV) for k in ("A", "B", "C", "D", "delta_bias"))
    res = ss.fwd(u, delta, A, B, C, D, z, bias, True)
    assert b == 1 or (res[0].stride() == delta.stride() and res[2].stride() == z.stride())
    o_out, o_outz, o_last = oracle.selective_scan_fwd(c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"],
                                                      c["delta_bias"], True)
    close(res[0], o_out, RTOL, ATOL, "out")
    close(res[2], o_outz, RTOL, ATOL, "out_z")
    close(res[1][:, :, -1, 1::2], o_last, RTOL, ATOL, "last_state")
    # dz given as a pre-allocated strided view (selective_scan_interface.py:244-251)
    dxz = torch.empty(2 * d, b, l, device=DEV).permute(1, 0, 2)
    dz_view = dxz[:, d:, :]
    r = ss.bwd(u, delta, A, B, C, D, z, bias, dout, res[1], res[0], dz_view, True, True)
    og = oracle.selective_scan_bwd(c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"], c["delta_bias"],
                                   c["dout"], True)
    assert r[7].data_ptr() == dz_view.data_ptr()
    close(r[0], og["du"], RTOL * 2, ATOL * 2, "du")
    close(r[1], og["ddelta"], RTOL * 5, ATOL * 10, "ddelta")
    close(r[2], og["dA"], RTOLW, ATOLW * 5 * max(1, l // 1024), "dA")
    close(r[3], og["dB"], RTOL, ATOL, "dB")
    close(r[4], og["dC"], RTOL, ATOL, "dC")
    close(r[5], og["dD"], RTOLW, ATOLW * max(1, l // 1024), "dD")
    close(r[6], og["ddelta_bias"], RTOLW, ATOLW * max(1, l // 1024), "ddelta_bias")
    close(r[7], og["dz"], RTOLW, ATOLW, "dz")
    close(r[8], o_outz, RTOL, ATOL, "out_z (recomputed)")


@pytest.mark.parametrize("case", [
    # (batch, dim, L, groups, dtype, layout, has_z, want_out)
    (4, 128, 1024, 1, torch.float32, "dbl", True, True),
    (8, 64, 2048, 1, torch.float32, "dbl", True, False),
    (1, 512, 1536, 1, torch.float32, "bdl", True, True),
    (2, 256, 512, 2, torch.float32, "bdl", False, True),
    (4, 128, 1024, 1, torch.bfloat16, "dbl", True, True),
])
def test_scan_fwd_stream_vs_oracle_and_chunk_path(case):
    """The streaming forward (selective_scan_stream.hip: taken when batch*dim >= 512, dstate 16, L % 512 == 0)
    against the CPU oracle, and against the chunk-parallel kernels on the same inputs (MMU_SCAN_STREAM=0):
    outputs, every chunk state the backward reads, and a backward pass fed with the streamed states."""
    import oracle
    from mm_unet_amd import selective_scan_hip as ss
    b, d, l, g, dt, layout, has_z, want_out = case
    n = 16
    gen = torch.Generator().manual_seed(21)
    c = _rand_case(b, d, l, n, seed=21)
    c["B"] = torch.randn(b, g, n, l, generator=gen)
    c["C"] = torch.randn(b, g, n, l, generator=gen)
    if dt == torch.bfloat16:
        c = {k: (v.bfloat16().float() if k in ("u", "delta", "z", "B", "C", "dout") else v) for k, v in c.items()}

    def put(t):
        t = t.to(dt)
        if layout == "dbl":
            return t.permute(1, 0, 2).contiguous().to(DEV).permute(1, 0, 2)
        return t.to(DEV)

    u, delta, z, dout = put(c["u"]), put(c["delta"]), put(c["z"]) if has_z else None, put(c["dout"])
    A, D, bias = (c[k].to(DEV) for k in ("A", "D", "delta_bias"))
    B, C = c["B"].to(dt).to(DEV), c["C"].to(dt).to(DEV)
    assert os.environ.get("MMU_SCAN_STREAM", "1") != "0"
    res = ss.fwd(u, delta, A, B, C, D, z, bias, True, want_out=want_out)
    os.environ["MMU_SCAN_STREAM"] = "0"
    try:
        ref = ss.fwd(u, delta, A, B, C, D, z, bias, True, want_out=want_out)
    finally:
        del os.environ["MMU_SCAN_STREAM"]
    torch.cuda.synchronize()
    # oracle (per group: the C oracle takes one group at a time)
    dpg = d // g
    o_out = torch.empty(b, d, l)
    o_outz = torch.empty(b, d, l)
    o_last = torch.empty(b, d, n)
    for gi in range(g):
        sl = slice(gi * dpg, (gi + 1) * dpg)
        oo, oz, ol = oracle.selective_scan_fwd(c["u"][:, sl], c["delta"][:, sl], c["A"][sl], c["B"][:, gi:gi + 1],
                                               c["C"][:, gi:gi + 1], c["D"][sl], c["z"][:, sl] if has_z else None,
                                               c["delta_bias"][sl], True)
        o_out[:, sl], o_last[:, sl] = oo, ol
        if has_z:
            o_outz[:, sl] = oz
    rt, at = (RTOL, ATOL) if dt == torch.float32 else (3e-2, 5e-2)
    if want_out:
        close(res[0], o_out, rt, at, "out vs oracle")
        close(res[0], ref[0], rt, at, "out vs chunk path")
    else:
        assert res[0] is None
    if has_z:
        close(res[2], o_outz, rt, at, "out_z vs oracle")
        close(res[2], ref[2], rt, at, "out_z vs chunk path")
    close(res[1][:, :, -1, 1::2], o_last, RTOL, ATOL, "last_state vs oracle")
    close(res[1][..., 1::2], ref[1][..., 1::2], RTOL, ATOL, "chunk states vs chunk path")
    if g == 1 and has_z and dt == torch.float32:
        r = ss.bwd(u, delta, A, B, C, D, z, bias, dout, res[1], None, None, True, False)
        og = oracle.selective_scan_bwd(c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"], c["delta_bias"],
                                       c["dout"], True)
        close(r[0], og["du"], RTOL * 2, ATOL * 2, "du (streamed states)")
        close(r[1], og["ddelta"], RTOL * 5, ATOL * 10, "ddelta (streamed states)")
        close(r[3], og["dB"], RTOL, ATOL * max(1, d // 32), "dB (streamed states)")
        close(r[4], og["dC"], RTOL, ATOL * max(1, d // 32), "dC (streamed states)")


@pytest.mark.parametrize("case", [
    # (batch, dim, L, groups, dtype, has_z, recompute_out_z, D and bias given)
    (2, 8, 1024, 1, torch.float32, True, True, True),     # two tiles: carry-in from the left / right neighbour
    (1, 6, 512, 1, torch.float32, True, False, True),     # one tile: no carries at all
    (2, 12, 1536, 2, torch.float32, True, True, True),    # two groups, three tiles (a middle tile with both carries)
    (2, 4, 1024, 1, torch.float32, False, False, False),  # no z, no D, no bias
    (1, 1, 2048, 1, torch.float32, True, False, True),    # a single channel: the pipeline prologue is the whole loop
    (2, 8, 1024, 1, torch.bfloat16, True, True, True),
    (2, 40, 1024, 1, torch.float32, True, False, True),   # few tiles, >= 16 channels: channel ranges (d_splits = 5)
])
def test_scan_bwd_w8_vs_oracle_and_p4(case):
    """chunk_apply_bwd_w8_kernel (selective_scan_bwd_w8.hip: 512-token tiles, one state pair per wave; forced with
    MMU_SCAN_BWD_W8=1, by default taken once the launch has a tile per CU) against the CPU oracle and against the
    256-token-tile kernel it replaces (MMU_SCAN_BWD_W8=0) on the same inputs and chunk states."""
    import oracle
    from mm_unet_amd import selective_scan_hip as ss
    b, d, l, g, dt, has_z, want_oz, has_db = case
    n = 16
    gen = torch.Generator().manual_seed(31)
    c = _rand_case(b, d, l, n, seed=31)
    c["B"] = torch.randn(b, g, n, l, generator=gen)
    c["C"] = torch.randn(b, g, n, l, generator=gen)
    if dt == torch.bfloat16:
        c = {k: (v.bfloat16().float() if k in ("u", "delta", "z", "B", "C", "dout") else v) for k, v in c.items()}

    def put(t):
        return t.to(dt).permute(1, 0, 2).contiguous().to(DEV).permute(1, 0, 2)

    u, delta, dout = put(c["u"]), put(c["delta"]), put(c["dout"])
    z = put(c["z"]) if has_z else None
    A = c["A"].to(DEV)
    D = c["D"].to(DEV) if has_db else None
    bias = c["delta_bias"].to(DEV) if has_db else None
    B, C = c["B"].to(dt).to(DEV), c["C"].to(dt).to(DEV)
    res = ss.fwd(u, delta, A, B, C, D, z, bias, True)
    got = {}
    for mode in ("1", "0"):
        os.environ["MMU_SCAN_BWD_W8"] = mode
        try:
            got[mode] = ss.bwd(u, delta, A, B, C, D, z, bias, dout, res[1], None, None, True, want_oz)
            torch.cuda.synchronize()
        finally:
            del os.environ["MMU_SCAN_BWD_W8"]
    # the same kernel fed with the forward's un-gated `out` (what the reference's backward is handed,
    # selective_scan.cpp:338): y is then read, not recomputed -- dz and the recomputed out_z must not move
    if has_z:
        os.environ["MMU_SCAN_BWD_W8"] = "1"
        try:
            got["y"] = ss.bwd(u, delta, A, B, C, D, z, bias, dout, res[1], res[0], None, True, want_oz)
            torch.cuda.synchronize()
        finally:
            del os.environ["MMU_SCAN_BWD_W8"]
    dpg = d // g
    og = {}
    for gi in range(g):
        sl = slice(gi * dpg, (gi + 1) * dpg)
        o = oracle.selective_scan_bwd(c["u"][:, sl], c["delta"][:, sl], c["A"][sl], c["B"][:, gi:gi + 1],
                                      c["C"][:, gi:gi + 1], c["D"][sl] if has_db else None,
                                      c["z"][:, sl] if has_z else None, c["delta_bias"][sl] if has_db else None,
                                      c["dout"][:, sl], True)
        for k, v in o.items():
            if v is not None:
                og.setdefault(k, []).append(v)
    cat = {k: torch.cat(v, dim=1 if k in ("du", "ddelta", "dz", "dB", "dC") else 0) for k, v in og.items()}
    f32 = dt == torch.float32
    rt, at = (RTOL, ATOL) if f32 else (3e-2, 6e-2)
    names = ["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz", "out_z"]
    w8, p4 = got["1"], got["0"]
    if has_z:
        wy = got["y"]
        assert len(wy) == len(w8)
        close(wy[7], cat["dz"], RTOLW if f32 else 3e-2, ATOLW if f32 else 6e-2, "dz vs oracle (y read)")
        for i, nm in enumerate(names[:len(w8)]):
            if w8[i] is None:
                assert wy[i] is None, nm
                continue
            scale = float(w8[i].float().abs().max()) + 1e-6
            # everything but dz / out_z is bit-identical (y enters nothing else); those two see y rounded to the I/O type
            tol = 0.0 if nm not in ("dz", "out_z") else (2e-5 if f32 else 2e-2) * scale
            close(wy[i], w8[i], 0.0, tol, f"{nm}: y read vs y recomputed")
    close(w8[0], cat["du"], rt * 2, at * 2, "du vs oracle")
    close(w8[1], cat["ddelta"], rt * 5, at * 10, "ddelta vs oracle")
    close(w8[2], cat["dA"], RTOLW if f32 else 3e-2, (ATOLW * 5 if f32 else 0.5) * max(1, l // 1024), "dA vs oracle")
    close(w8[3], cat["dB"], rt, at, "dB vs oracle")
    close(w8[4], cat["dC"], rt, at, "dC vs oracle")
    if has_db:
        close(w8[5], cat["dD"], RTOLW if f32 else 3e-2, (ATOLW if f32 else 0.5) * max(1, l // 1024), "dD vs oracle")
        close(w8[6], cat["ddelta_bias"], RTOLW if f32 else 3e-2, (ATOLW if f32 else 0.5) * max(1, l // 1024), "dbias vs oracle")
    if has_z:
        close(w8[7], cat["dz"], RTOLW if f32 else 3e-2, ATOLW if f32 else 6e-2, "dz vs oracle")
    # the two kernels share every input and all the math: fp32 summation order is the only difference
    assert len(w8) == len(p4)
    for i, nm in enumerate(names[:len(w8)]):
        if w8[i] is None or p4[i] is None:
            assert w8[i] is None and p4[i] is None, nm
            continue
        scale = float(p4[i].float().abs().max()) + 1e-6
        close(w8[i], p4[i], 0.0, (2e-5 if f32 else 2e-2) * scale, f"{nm} vs the 256-token-tile kernel")


def test_scan_state_groups_dstate64_vs_golden_and_generic(monkeypatch):
    """dstate = 64 (BASELINE config 5) as four dstate-16 launches on strided views (selective_scan_hip._fwd_groups /
    _bwd_groups) against the reference fixture ``scan_c5_D8_L512_N64`` and against the single generic-dstate launch on a
    larger case: outputs, every chunk state, every gradient."""
    from mm_unet_amd import selective_scan_hip as ss
    monkeypatch.setattr(ss, "GROUP_SPLIT_MIN_ELEMENTS", 1)
    g = golden("scan_c5_D8_L512_N64")
    u, delta, A, B, C = (_g(g, k) for k in ("u", "delta", "A", "B", "C"))
    D, z, bias, dout = (_g(g, k) for k in ("D", "z", "delta_bias", "dout"))
    sp = bool(g["softplus"])
    assert ss.group_split(A.shape[1], u)
    res = ss.fwd(u, delta, A, _bc4(B), _bc4(C), D, z, bias, sp)
    close(res[2] if z is not None else res[0], g["out"], RTOL, ATOL, "out")
    close(res[1][:, :, -1, 1::2], g["last_state"], RTOL, ATOL, "last_state")
    r = ss.bwd(u, delta, A, _bc4(B), _bc4(C), D, z, bias, dout, res[1], res[0], None, sp, z is not None)
    close(r[0], g["du"], RTOL * 2, ATOL * 2, "du")
    close(r[1], g["ddelta"], RTOL * 5, ATOL * 10, "ddelta")
    close(r[2], g["dA"], RTOLW, ATOLW * 5, "dA")
    close(r[3].reshape(g["dB"].shape), g["dB"], RTOL, ATOL, "dB")
    close(r[4].reshape(g["dC"].shape), g["dC"], RTOL, ATOL, "dC")
    for i, k in ((5, "dD"), (6, "ddelta_bias"), (7, "dz")):
        if k in g:
            close(r[i], g[k], RTOLW, ATOLW, k)
    # larger case, mamba layout, against the generic single launch (and the bf16 contract)
    for dt, rt, at in ((torch.float32, RTOL, ATOL), (torch.bfloat16, 3e-2, 6e-2)):
        c = _rand_case(2, 16, 2048, 64, seed=4)
        mk = lambda t_: t_.to(dt).permute(1, 0, 2).contiguous().to(DEV).permute(1, 0, 2)  # noqa: E731
        uu, dd, zz, go = mk(c["u"]), mk(c["delta"]), mk(c["z"]), mk(c["dout"])
        AA, DD, bb = c["A"].to(DEV), c["D"].to(DEV), c["delta_bias"].to(DEV)
        BB, CC = c["B"].to(dt).to(DEV), c["C"].to(dt).to(DEV)
        a = ss.fwd(uu, dd, AA, BB, CC, DD, zz, bb, True)
        ga = ss.bwd(uu, dd, AA, BB, CC, DD, zz, bb, go, a[1], a[0], None, True, True)
        monkeypatch.setattr(ss, "GROUP_SPLIT", False)
        b_ = ss.fwd(uu, dd, AA, BB, CC, DD, zz, bb, True)
        gb = ss.bwd(uu, dd, AA, BB, CC, DD, zz, bb, go, b_[1], b_[0], None, True, True)
        monkeypatch.setattr(ss, "GROUP_SPLIT", True)
        close(a[0], b_[0], rt, at, f"out {dt}")
        close(a[2], b_[2], rt, at, f"out_z {dt}")
        close(a[1][..., 1::2], b_[1][..., 1::2], RTOL, ATOL, f"chunk states {dt}")
        for i, nm in enumerate(["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz", "out_z"]):
            sc = max(1.0, float(gb[i].float().abs().max()))
            close(ga[i], gb[i], rt * 3, at * 3 * (sc if nm in ("dA", "dD", "ddelta_bias") else 1.0), f"{nm} {dt}")


@pytest.mark.parametrize("dstate", [32, 128])
def test_scan_large_dstate_without_equal_chunk_length_stays_on_generic_kernels(monkeypatch, dstate):
    """d_state 32 and 80..128 have a different chunk length from the dstate-16 kernels (256 / 64 vs 128), so
    they cannot run as state groups; group_split() must say so and fwd / bwd must take the generic kernels instead of
    raising 'group split needs equal chunk lengths'.  Checked against the oracle."""
    import oracle
    from mm_unet_amd import selective_scan_hip as ss
    monkeypatch.setattr(ss, "GROUP_SPLIT_MIN_ELEMENTS", 1)
    c = _rand_case(1, 4, 1024, dstate, seed=9, device="cpu")
    assert not ss.group_split(dstate, c["u"].to(DEV))
    d = {k: v.to(DEV) for k, v in c.items()}
    res = ss.fwd(d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["delta_bias"], True)
    _, ref_z, ref_last = oracle.selective_scan_fwd(c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"],
                                                   c["delta_bias"], True)
    close(res[2], ref_z, RTOL, ATOL, "out_z")
    close(res[1][:, :, -1, 1::2], ref_last, RTOL, ATOL, "last_state")
    r = ss.bwd(d["u"], d["delta"], d["A"], d["B"], d["C"], d["D"], d["z"], d["delta_bias"], d["dout"], res[1], res[0],
               None, True, False)
    ref = oracle.selective_scan_bwd(c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"], c["delta_bias"],
                                    c["dout"], True)
    close(r[0], ref["du"], RTOL * 2, ATOL * 2, "du")
    close(r[1], ref["ddelta"], RTOL * 5, ATOL * 10, "ddelta")
    close(r[3].reshape(ref["dB"].shape), ref["dB"], RTOL, ATOL, "dB")
    close(r[4].reshape(ref["dC"].shape), ref["dC"], RTOL, ATOL, "dC")


def test_scan_bwd_reproducibility():
    """Every gradient of the dstate-16 backward (K4p / K4s: register dB/dC sums, no atomics anywhere) is
    bit-identical run to run; the reference uses global float atomics for dA/dB/dC/dD/dbias
    (selective_scan_bwd_kernel.cuh:312-313,469-487).  (Only the generic-dstate fallback still sums dB/dC
    with LDS float atomics.)"""
    from mm_unet_amd import selective_scan_hip as ss
    c = {k: v.to(DEV) for k, v in _rand_case(2, 6, 2048, 16, seed=5).items()}
    res = ss.fwd(c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"], c["delta_bias"], True)
    outs = []
    for _ in range(3):
        r = ss.bwd(c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"], c["delta_bias"], c["dout"], res[1],
                   res[0], None, True, False)
        outs.append([t.clone() for t in r[:8]])
    names = ["du", "ddelta", "dA", "dB", "dC", "dD", "ddelta_bias", "dz"]
    for other in outs[1:]:
        for nm, a, b_ in zip(names, outs[0], other):
            assert torch.equal(a, b_), f"{nm} is not run-to-run bit-identical"


def test_scan_full_size_properties():
    """BASELINE headline shape (B=8, D=128, L=65536, N=16): causality (a prefix of the output
    depends only on the prefix of the input), linearity in u, and a 3-channel spot check
    against the oracle."""
    import oracle
    from mm_unet_amd import selective_scan_hip as ss
    b, d, l, n = 8, 128, 65536, 16
    gen = torch.Generator(device=DEV).manual_seed(0)
    A = -0.5 * torch.rand(d, n, device=DEV, generator=gen)
    B = torch.randn(b, 1, n, l, device=DEV, generator=gen)
    C = torch.randn(b, 1, n, l, device=DEV, generator=gen)
    D = torch.randn(d, device=DEV, generator=gen)
    z = torch.randn(b, d, l, device=DEV, generator=gen)
    bias = 0.5 * torch.rand(d, device=DEV, generator=gen)
    u = torch.randn(b, d, l, device=DEV, generator=gen)
    delta = 0.5 * torch.rand(b, d, l, device=DEV, generator=gen)
    out, x, out_z = ss.fwd(u, delta, A, B, C, D, z, bias, True)
    # causality
    lp = 4096 + 256
    out_p, _, out_z_p = ss.fwd(u[..., :lp].contiguous(), delta[..., :lp].contiguous(), A,
                               B[..., :lp].contiguous(), C[..., :lp].contiguous(), D, z[..., :lp].contiguous(), bias,
                               True)
    # (the ragged prefix takes the generic 128-token-tile kernel, the full length the packed 512-token one:
    # same recurrence, different summation order over the states -> fp32 rounding-level differences)
    close(out[..., :lp], out_p, 1e-4, 1e-3, "causality(out)")
    close(out_z[..., :lp], out_z_p, 1e-4, 1e-3, "causality(out_z)")
    # linearity in u (delta fixed): out(2u) == 2 out(u)
    out2, _, _ = ss.fwd(2 * u, delta, A, B, C, D, z, bias, True)
    close(out2, 2 * out, 1e-4, 1e-4, "linearity")
    # oracle spot check on batch 3, channels 0, 77, 127
    sel = [0, 77, 127]
    o_out, o_outz, o_last = oracle.selective_scan_fwd(u[3:4, sel].cpu(), delta[3:4, sel].cpu(), A[sel].cpu(),
                                                      B[3:4].cpu(), C[3:4].cpu(), D[sel].cpu(), z[3:4, sel].cpu(),
                                                      bias[sel].cpu(), True)
    close(out_z[3:4, sel], o_outz, RTOL, ATOL, "out_z vs oracle")
    close(x[3:4, sel, -1, 1::2], o_last, RTOL, ATOL, "last_state vs oracle")


@pytest.mark.parametrize("shape", [(2, 6, 1024, 16), (1, 128, 1024, 16)])
def test_scan_bf16_io(shape):
    """bf16 I/O contract: u, delta, z, B, C, out in bf16; A, D, bias, states fp32."""
    import oracle
    from mm_unet_amd import selective_scan_hip as ss
    b, d, l, n = shape
    c = _rand_case(b, d, l, n, seed=2)
    q = {k: (v.bfloat16() if k in ("u", "delta", "z", "B", "C", "dout") else v) for k, v in c.items()}
    g = {k: v.to(DEV) for k, v in q.items()}
    res = ss.fwd(g["u"], g["delta"], g["A"], g["B"], g["C"], g["D"], g["z"], g["delta_bias"], True)
    assert res[2].dtype == torch.bfloat16 and res[1].dtype == torch.float32
    f = {k: v.float() for k, v in q.items()}
    _, o_outz, _ = oracle.selective_scan_fwd(f["u"], f["delta"], f["A"], f["B"], f["C"], f["D"], f["z"],
                                             f["delta_bias"], True)
    close(res[2], o_outz, 3e-2, 5e-2, "out_z (bf16)")
    r = ss.bwd(g["u"], g["delta"], g["A"], g["B"], g["C"], g["D"], g["z"], g["delta_bias"], g["dout"], res[1],
               res[0], None, True, False)
    og = oracle.selective_scan_bwd(f["u"], f["delta"], f["A"], f["B"], f["C"], f["D"], f["z"], f["delta_bias"],
                                   f["dout"], True)
    close(r[0], og["du"], 6e-2, 1e-1, "du (bf16)")
    close(r[2], og["dA"], 3e-2, 5e-1, "dA (bf16)")
    assert r[3].dtype == torch.bfloat16


def test_scan_rejects_bad_arguments():
    from mm_unet_amd import selective_scan_hip as ss
    c = {k: v.to(DEV) for k, v in _rand_case(1, 4, 64, 16, seed=1).items()}
    with pytest.raises(RuntimeError):
        ss.fwd(c["u"].cpu(), c["delta"], c["A"], c["B"], c["C"], None, None, None, False)
    with pytest.raises(RuntimeError):
        ss.fwd(c["u"], c["delta"][..., :32], c["A"], c["B"], c["C"], None, None, None, False)
    with pytest.raises(RuntimeError):
        ss.fwd(c["u"].transpose(1, 2).contiguous().transpose(1, 2), c["delta"], c["A"], c["B"], c["C"], None, None,
               None, False)
    with pytest.raises(RuntimeError):  # dstate > 128
        ss.fwd(c["u"], c["delta"], torch.zeros(4, 200, device=DEV), torch.zeros(1, 1, 200, 64, device=DEV),
               torch.zeros(1, 1, 200, 64, device=DEV), None, None, None, False)


# --------------------------------------------------------------------------- autograd wrappers (SSI:14-83, CCI:10-46)
@pytest.mark.parametrize("name", SCAN)
def test_selective_scan_fn_autograd_golden(name):
    """Backpropagates through the public ``selective_scan_fn`` (SelectiveScanFn.backward), with and without
    ``return_last_state``, against the reference's selective_scan_ref fixtures."""
    from mm_unet_amd.selective_scan_interface import selective_scan_fn
    g = golden(name)
    leaves = {k: _g(g, k) for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias")}
    for v in leaves.values():
        if v is not None:
            v.requires_grad_()
    sp = bool(g["softplus"])
    out, last = selective_scan_fn(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"],
                                  leaves["z"], leaves["delta_bias"], sp, return_last_state=True)
    close(out, g["out"], RTOL, ATOL, "out")
    close(last, g["last_state"], RTOL, ATOL, "last_state")
    assert not last.requires_grad or last.grad_fn is not None
    out.backward(_g(g, "dout"))
    close(leaves["u"].grad, g["du"], RTOL * 2, ATOL * 2, "du")
    close(leaves["delta"].grad, g["ddelta"], RTOL * 5, ATOL * 10, "ddelta")
    close(leaves["A"].grad, g["dA"], RTOLW, ATOLW * 5, "dA")
    close(leaves["B"].grad, g["dB"], RTOL, ATOL, "dB")
    close(leaves["C"].grad, g["dC"], RTOL, ATOL, "dC")
    for k, gk in (("D", "dD"), ("delta_bias", "ddelta_bias"), ("z", "dz")):
        if gk in g:
            close(leaves[k].grad, g[gk], RTOLW, ATOLW, gk)
    # plain call (no last state) gives the same output object semantics
    out2 = selective_scan_fn(*(v.detach() if v is not None else None for v in
                               (leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"],
                                leaves["z"], leaves["delta_bias"])), sp)
    assert torch.equal(out2, out.detach())


@pytest.mark.parametrize("name", CONV)
def test_causal_conv1d_fn_autograd_golden(name):
    """Backpropagates through the public ``causal_conv1d_fn`` against the reference's causal_conv1d_ref fixtures."""
    from mm_unet_amd.causal_conv1d_interface import causal_conv1d_fn
    g = golden(name)
    x, w, b = _g(g, "x").requires_grad_(), _g(g, "weight").requires_grad_(), _g(g, "bias")
    if b is not None:
        b.requires_grad_()
    out = causal_conv1d_fn(x, w, b, "silu" if bool(g["silu"]) else None)
    close(out, g["out"], 3e-4, 1e-3, "out")
    out.backward(_g(g, "dout"))
    close(x.grad, g["dx"], 3e-4, 1e-3, "dx")
    close(w.grad, g["dweight"], 1e-3, 1e-3, "dweight")
    if "dbias" in g:
        close(b.grad, g["dbias"], 1e-3, 1e-3, "dbias")


def test_bimamba_inner_fn_vs_oracle_composition():
    """``bimamba_inner_fn`` (exported by the reference, SSI:616-624; not called by MM-UNet) forward + every
    gradient against bimamba_inner_ref's steps (SSI:673-709) composed from the CPU oracle's differentiable
    conv1d / scan and torch CPU linear algebra."""
    import oracle
    import torch.nn.functional as F
    from mm_unet_amd.selective_scan_interface import BiMambaInnerFn, bimamba_inner_fn
    gen = torch.Generator().manual_seed(33)
    b, d, l, n, r, e = 2, 8, 192, 16, 2, 4

    def rnd(*s, scale=1.0):
        return (scale * torch.randn(*s, generator=gen))

    P = dict(xz=rnd(b, 2 * d, l), conv_w=rnd(d, 1, 4, scale=0.5), conv_b=rnd(d, scale=0.1),
             x_proj=rnd(r + 2 * n, d, scale=0.3), dt_proj=rnd(d, r, scale=0.5), out_proj=rnd(e, d, scale=0.3),
             out_bias=rnd(e, scale=0.1), A=-torch.rand(d, n, generator=gen) - 0.1,
             A_b=-torch.rand(d, n, generator=gen) - 0.1, D=rnd(d), dt_bias=0.5 * torch.rand(d, generator=gen))
    dout = rnd(b, l, e)

    def ref(p):
        x, z = p["xz"].chunk(2, dim=1)
        x = oracle.causal_conv1d(x.contiguous(), p["conv_w"].view(d, 4), p["conv_b"], "silu")
        x_dbl = F.linear(x.permute(0, 2, 1).reshape(b * l, d), p["x_proj"])
        delta = (p["dt_proj"] @ x_dbl[:, :r].t()).view(d, b, l).permute(1, 0, 2).contiguous()
        B = x_dbl[:, r:r + n].view(b, l, n).permute(0, 2, 1).contiguous()
        C = x_dbl[:, -n:].view(b, l, n).permute(0, 2, 1).contiguous()
        z = z.contiguous()
        y = oracle.selective_scan(x, delta, p["A"], B, C, p["D"], z, p["dt_bias"], True)
        y_b = oracle.selective_scan(x.flip([-1]).contiguous(), delta.flip([-1]).contiguous(), p["A_b"],
                                    B.flip([-1]).contiguous(), C.flip([-1]).contiguous(), p["D"],
                                    z.flip([-1]).contiguous(), p["dt_bias"], True)
        y = y + y_b.flip([-1])
        return F.linear(y.permute(0, 2, 1), p["out_proj"], p["out_bias"])

    pc = {k: v.clone().requires_grad_() for k, v in P.items()}
    o_ref = ref(pc)
    o_ref.backward(dout)
    pg = {k: v.clone().to(DEV).requires_grad_() for k, v in P.items()}
    args = (pg["xz"], pg["conv_w"], pg["conv_b"], pg["x_proj"], pg["dt_proj"], pg["out_proj"], pg["out_bias"],
            pg["A"], pg["A_b"], None, None, pg["D"], pg["dt_bias"], None, None, True)
    out = bimamba_inner_fn(*args)
    close(out, o_ref, RTOL, ATOL, "out")
    assert torch.equal(BiMambaInnerFn.apply(*args), out)
    out.backward(dout.to(DEV))
    for k in P:
        scale = float(pc[k].grad.abs().max())
        close(pg[k].grad, pc[k].grad, 2e-3, 2e-3 * max(scale, 1.0), f"d{k}")


# --------------------------------------------------------------------------- causal conv1d
@pytest.mark.parametrize("name", CONV)
def test_conv1d_golden(name):
    from mm_unet_amd import causal_conv1d_hip as cc
    g = golden(name)
    silu = bool(g["silu"])
    x, w, b, dout = _g(g, "x"), _g(g, "weight"), _g(g, "bias"), _g(g, "dout")
    out = cc.causal_conv1d_fwd(x, w, b, silu)
    close(out, g["out"], 3e-4, 1e-3, "out")
    dx, dw, db = cc.causal_conv1d_bwd(x, w, b, dout, None, silu)
    close(dx, g["dx"], 3e-4, 1e-3, "dx")
    close(dw, g["dweight"], 1e-3, 1e-3, "dweight")
    if "dbias" in g:
        close(db, g["dbias"], 1e-3, 1e-3, "dbias")


@pytest.mark.parametrize("shape", [(2, 6, 4099), (2, 12, 1024), (1, 256, 2048), (3, 4, 3)])
def test_conv1d_vs_oracle_strided(shape):
    """x is the first half of an xz tensor laid out [2D][B][L] (selective_scan_interface.py:175);
    dx is written into a view of dxz (:244-245,281-283)."""
    import oracle
    from mm_unet_amd import causal_conv1d_hip as cc
    b, d, l = shape
    gen = torch.Generator().manual_seed(7)
    xz = torch.randn(2 * d, b, l, generator=gen)
    w = torch.randn(d, 4, generator=gen)
    bias = torch.randn(d, generator=gen)
    dout = torch.randn(b, d, l, generator=gen)
    xz_g = xz.to(DEV).permute(1, 0, 2)
    x_g = xz_g[:, :d]
    out = cc.causal_conv1d_fwd(x_g, w.to(DEV), bias.to(DEV), True)
    x_c = xz.permute(1, 0, 2)[:, :d].contiguous()
    close(out, oracle.causal_conv1d_fwd(x_c, w, bias, True), 3e-4, 1e-3, "out")
    dxz = torch.zeros(2 * d, b, l, device=DEV).permute(1, 0, 2)
    dx_view = dxz[:, :d]
    dx, dw, db = cc.causal_conv1d_bwd(x_g, w.to(DEV), bias.to(DEV), dout.to(DEV), dx_view, True)
    odx, odw, odb = oracle.causal_conv1d_bwd(x_c, w, bias, dout, True)
    assert dx.data_ptr() == dx_view.data_ptr()
    close(dx, odx, 3e-4, 1e-3, "dx")
    close(dw, odw, 1e-3, 1e-3 * max(1, l // 512), "dweight")
    close(db, odb, 1e-3, 1e-3 * max(1, l // 512), "dbias")
    assert float(dxz[:, d:].abs().max()) == 0.0, "conv1d bwd wrote outside its dx view"


def test_conv1d_bwd_weight_gradient_is_bit_reproducible_and_matches_atomic_path():
    """dW / db are per-block partial sums added in a fixed order (workspace path): two calls give identical bits, at
    a size where a block walks several tiles and a channel has several blocks.  The C-ABI's NULL-workspace path
    (float atomics into zeroed buffers, the reference's scheme, causal_conv1d_bwd.cu:256-268) gives the same sums
    to rounding."""
    from mm_unet_amd import _lib
    from mm_unet_amd import causal_conv1d_hip as cc
    gen = torch.Generator().manual_seed(11)
    for (b, d, l, width) in [(2, 24, 65536 + 37, 4), (1, 8, 5000, 3), (8, 768, 1024, 4)]:
        x = torch.randn(b, d, l, generator=gen).to(DEV)
        w = torch.randn(d, width, generator=gen).to(DEV)
        bias = torch.randn(d, generator=gen).to(DEV)
        dout = torch.randn(b, d, l, generator=gen).to(DEV)
        dx1, dw1, db1 = cc.causal_conv1d_bwd(x, w, bias, dout, None, True)
        dx2, dw2, db2 = cc.causal_conv1d_bwd(x, w, bias, dout, None, True)
        assert torch.equal(dw1, dw2) and torch.equal(db1, db2) and torch.equal(dx1, dx2)
        # the atomics path of the ABI
        dwa = torch.zeros(d, width, device=DEV)
        dba = torch.zeros(d, device=DEV)
        dxa = torch.empty_like(x)
        p = _lib.Conv1dBwdParams()
        p.batch, p.dim, p.seqlen, p.width = b, d, l, width
        p.dtype, p.silu = _lib.dtype_code(x), 1
        p.x, p.weight, p.bias = x.data_ptr(), w.data_ptr(), bias.data_ptr()
        p.dout, p.dx, p.dweight, p.dbias = dout.data_ptr(), dxa.data_ptr(), dwa.data_ptr(), dba.data_ptr()
        p.x_bs, p.x_ds = x.stride(0), x.stride(1)
        p.dout_bs, p.dout_ds = dout.stride(0), dout.stride(1)
        p.dx_bs, p.dx_ds = dxa.stride(0), dxa.stride(1)
        p.w_ds, p.w_ws = w.stride(0), w.stride(1)
        p.workspace = None
        _lib.check(_lib.lib().mmu_causal_conv1d_bwd(p, _lib.stream_of(x)))
        assert torch.equal(dxa, dx1)
        scale = float(dw1.abs().max())
        close(dwa, dw1, 1e-4, 1e-4 * scale, "dweight atomics vs ordered")
        close(dba, db1, 1e-4, 1e-4 * float(db1.abs().max()), "dbias atomics vs ordered")


def test_conv1d_update_matches_reference_semantics():
    """causal_conv1d_update_ref (causal_conv1d_interface.py:83-104): roll, append, dot, silu."""
    from mm_unet_amd import causal_conv1d_hip as cc
    gen = torch.Generator().manual_seed(9)
    b, d, w = 3, 70, 4
    x = torch.randn(b, d, generator=gen)
    state = torch.randn(b, d, w, generator=gen)
    weight = torch.randn(d, w, generator=gen)
    bias = torch.randn(d, generator=gen)
    st = state.clone()
    st = torch.roll(st, shifts=-1, dims=-1)
    st[:, :, -1] = x
    ref = torch.nn.functional.silu((st * weight).sum(-1) + bias)
    sg = state.to(DEV)
    out = cc.causal_conv1d_update(x.to(DEV), sg, weight.to(DEV), bias.to(DEV), True)
    close(out, ref, 1e-4, 1e-4, "out")
    assert torch.equal(sg.cpu(), st), "conv_state not rolled exactly"


def test_conv1d_full_size_shift_property():
    """(B=8, D=128, L=65536): causal conv of a sequence delayed by 4 tokens equals the delayed
    output (time invariance), and bias-free/no-activation conv is linear."""
    from mm_unet_amd import causal_conv1d_hip as cc
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(8, 128, 65536, device=DEV, generator=gen)
    w = torch.randn(128, 4, device=DEV, generator=gen)
    out = cc.causal_conv1d_fwd(x, w, None, False)
    xs = torch.zeros_like(x)
    xs[..., 4:] = x[..., :-4]
    outs = cc.causal_conv1d_fwd(xs, w, None, False)
    close(outs[..., 4:], out[..., :-4], 1e-6, 1e-6, "shift invariance")
    close(cc.causal_conv1d_fwd(3 * x, w, None, False), 3 * out, 1e-5, 1e-5, "linearity")


@pytest.mark.parametrize("case", [
    # (m, n, batch, L, a layout, b layout)   layouts: "cm" = [C][B][L] storage, "bm" = [B][C][L]
    (256, 64, 2, 1024, "cm", "bm"),       # in_proj weight gradient: G channel-major, X batch-major
    (64, 128, 3, 512, "bm", "cm"),        # out_proj
    (36, 128, 2, 2048, "cm", "cm"),       # x_proj: 36 rows (a partial 128-row tile)
    (128, 4, 2, 1024, "cm", "cm"),        # dt_proj: 4 columns (a partial 64-column tile)
    (64, 192, 1, 4096, "cm", "cm"),       # DSC weight gradient
    (200, 70, 2, 96, "bm", "bm"),         # ragged tile counts, three chunks per batch item
    (130, 65, 3, 384, "cm", "bm"),        # ragged tiles on the 128-token-step kernel, three steps per batch item
])
def test_gemm_nt_splitk_vs_float64(case):
    """csrc/gemm_nt_splitk.hip (token-contraction product on the matrix cores, operands addressed in place in either
    layout) against a float64 einsum, for the hi/lo-split bf16 form and the exact fp32 form; and bit-reproducible."""
    from mm_unet_amd import mfma_gemm
    m, n, b, l, la, lb = case
    gen = torch.Generator().manual_seed(17)

    def make(rows, layout):
        t = torch.randn(b, rows, l, generator=gen)
        if layout == "cm":
            st = t.permute(1, 0, 2).contiguous().to(DEV)        # storage [rows][B][L]
            return t, st, b * l, l
        return t, t.to(DEV), l, rows * l                        # storage [B][rows][L]

    a_ref, a_dev, a_rs, a_bs = make(m, la)
    b_ref, b_dev, b_rs, b_bs = make(n, lb)
    assert mfma_gemm.nt_supported(a_dev, b_dev, l)
    ref = torch.einsum("bil,bjl->ij", a_ref.double(), b_ref.double())
    # randn operands: a sum of T products has standard deviation sqrt(T); float32 accumulation rounds at 2^-24 of it per
    # add, the split form adds 2^-16-relative product errors that average out over T
    # three kernels: 128-token steps (taken when L % 128 == 0), 32-token steps, 32-token steps with exact products
    for exact, narrow, tol in ((False, False, 2e-5), (False, True, 2e-5), (True, True, 4e-6)):
        c1 = mfma_gemm.gemm_nt(a_dev, b_dev, m, n, b, l, a_rs, a_bs, b_rs, b_bs, exact=exact, narrow=narrow)
        c2 = mfma_gemm.gemm_nt(a_dev, b_dev, m, n, b, l, a_rs, a_bs, b_rs, b_bs, exact=exact, narrow=narrow)
        assert torch.equal(c1, c2)
        err = float((c1.double().cpu() - ref).abs().max())
        assert err <= tol * (b * l) ** 0.5, f"exact={exact} narrow={narrow}: max abs err {err:.3e}"


@pytest.mark.parametrize("case", [(256, 64, 2, 1024, "cm", "bm"), (64, 128, 3, 512, "bm", "cm"), (36, 128, 2, 2048, "cm", "cm"),
                                  (128, 4, 2, 1024, "cm", "cm"), (130, 65, 3, 384, "cm", "bm")])
def test_gemm_nt_bfloat16_operands_vs_float64(case):
    """gemm_nt with bfloat16 operands (autocast: gemm_nt_wide_kernel<true>, both operands exact bf16 values, ONE MFMA
    per product, float32 sums) against a float64 einsum of the SAME bf16 values: float32-grade, and reproducible."""
    from mm_unet_amd import mfma_gemm
    m, n, b, l, la, lb = case
    gen = torch.Generator().manual_seed(23)

    def make(rows, layout):
        t = torch.randn(b, rows, l, generator=gen).to(torch.bfloat16)
        if layout == "cm":
            return t, t.permute(1, 0, 2).contiguous().to(DEV), b * l, l
        return t, t.to(DEV), l, rows * l

    a_ref, a_dev, a_rs, a_bs = make(m, la)
    b_ref, b_dev, b_rs, b_bs = make(n, lb)
    assert mfma_gemm.nt_supported(a_dev, b_dev, l)
    ref = torch.einsum("bil,bjl->ij", a_ref.double(), b_ref.double())
    c1 = mfma_gemm.gemm_nt(a_dev, b_dev, m, n, b, l, a_rs, a_bs, b_rs, b_bs)
    c2 = mfma_gemm.gemm_nt(a_dev, b_dev, m, n, b, l, a_rs, a_bs, b_rs, b_bs)
    assert c1.dtype == torch.float32 and torch.equal(c1, c2)
    err = float((c1.double().cpu() - ref).abs().max())
    assert err <= 4e-6 * (b * l) ** 0.5, err       # exact products: only the float32 accumulation rounds
    with pytest.raises(RuntimeError):              # 32-token steps have no bf16 form
        mfma_gemm.gemm_nt(a_dev, b_dev, m, n, b, 96, a_rs, a_bs, b_rs, b_bs)


@pytest.mark.parametrize("shape", [(256, 64), (36, 128), (64, 192)])
def test_gemm_nt_full_size_properties(shape):
    """csrc/gemm_nt_splitk.hip at the headline token count (B * L = 8 * 65,536, the slab plans of the real calls): against
    the library's batched split-K on the same operands, batch-major vs channel-major addressing of the same data
    (identical bits: same contraction order), and linearity in the first operand."""
    from mm_unet_amd import mfma_gemm
    from mm_unet_amd.tall_gemm import nt_splitk
    m, n = shape
    B, L = 8, 65536
    gen = torch.Generator(device=DEV).manual_seed(3)
    a = torch.randn(m, B, L, device=DEV, generator=gen)          # storage [m][B][L]
    b = torch.randn(n, B, L, device=DEV, generator=gen)
    c = mfma_gemm.gemm_nt(a, b, m, n, B, L, B * L, L, B * L, L)
    mfma_gemm.NT_ENABLED = False
    try:
        ref = nt_splitk(a.view(m, B * L), b.view(n, B * L))
    finally:
        mfma_gemm.NT_ENABLED = True
    scale = (B * L) ** 0.5
    assert float((c - ref).abs().max()) <= 3e-5 * scale
    a_bm = a.permute(1, 0, 2).contiguous()                        # the same values stored [B][m][L]
    c_bm = mfma_gemm.gemm_nt(a_bm, b, m, n, B, L, L, m * L, B * L, L)
    assert torch.equal(c_bm, c)
    c2 = mfma_gemm.gemm_nt(2.0 * a, b, m, n, B, L, B * L, L, B * L, L)
    assert torch.equal(c2, 2.0 * c)                               # powers of two commute with every rounding


@pytest.mark.parametrize("case", [(2, 64, 64, 32, 64, 4), (1, 64, 128, 16, 24, 3), (2, 16, 64, 20, 12, 3),
                                  (1, 128, 256, 64, 64, 3), (3, 64, 64, 10, 130, 4)])
def test_conv_s2_mfma_vs_conv2d_fp64(case):
    """csrc/conv_s2_mfma.hip, strided form: Conv2d(k, stride 2, padding 1) (MMUNet.py:375,439) forward, input gradient
    (the transposed kernel) and weight gradient against ATen in float64; hi/lo bf16 split: ~2^-16 per product."""
    import torch.nn.functional as F
    from mm_unet_amd import conv_s2
    B, cin, cout, H, W, k = case
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, cin, H, W, generator=gen).to(DEV).requires_grad_()
    w = (torch.randn(cout, cin, k, k, generator=gen) / (k * cin ** 0.5)).to(DEV).requires_grad_()
    b = torch.randn(cout, generator=gen).to(DEV).requires_grad_()
    assert conv_s2.conv_supported(x, w)
    out = conv_s2.conv_s2(x, w, b)
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    ref = F.conv2d(xd, wd, bd, stride=2, padding=1)
    assert out.shape == ref.shape
    g = torch.randn(out.shape, generator=gen).to(DEV)
    out.backward(g)
    ref.backward(g.double())
    sc = lambda t: max(1.0, float(t.abs().max()))   # noqa: E731
    assert float((out.double() - ref).abs().max()) < 1e-4 * sc(ref), "forward"
    assert float((x.grad.double() - xd.grad).abs().max()) < 1e-4 * sc(xd.grad), "d input"
    assert float((w.grad.double() - wd.grad).abs().max()) < 2e-4 * sc(wd.grad), "d weight"
    assert float((b.grad.double() - bd.grad).abs().max()) < 1e-4 * sc(bd.grad), "d bias"


@pytest.mark.parametrize("case", [("conv", 2, 64, 64, 32, 64, 4), ("conv", 1, 64, 128, 16, 24, 3), ("conv", 3, 64, 64, 10, 130, 4),
                                  ("convt", 2, 64, 64, 16, 32, 4), ("convt", 1, 128, 128, 6, 70, 4)])
def test_conv_s2_bfloat16_activations_vs_fp64(case):
    """csrc/conv_s2_mfma.hip under bf16 autocast (XB forms of the producer / consumer kernel: bf16 input / output, float32
    weights, two MFMAs per product; weight gradient: both operands exact bf16, ONE MFMA per product): strided and
    transposed convolution, output and input gradient within bf16 rounding of float64 on the SAME bf16 inputs, weight
    and bias gradient to float32 accuracy."""
    import torch.nn.functional as F
    from mm_unet_amd import conv_s2
    kind, B, cin, cout, H, W, k = case
    gen = torch.Generator().manual_seed(B + cin + H)
    x = torch.randn(B, cin, H, W, generator=gen).to(torch.bfloat16)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    w = torch.randn(wshape, generator=gen) / (k * cin ** 0.5)
    b = torch.randn(cout, generator=gen)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    ref = F.conv2d(xr, wr, br, stride=2, padding=1) if kind == "conv" else F.conv_transpose2d(xr, wr, br, stride=2, padding=1)
    g = torch.randn(ref.shape, generator=gen).to(torch.bfloat16)
    ref.backward(g.double())
    xg, wg, bg = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert (conv_s2.conv_supported if kind == "conv" else conv_s2.convt_supported)(xg, wg)
        out = (conv_s2.conv_s2 if kind == "conv" else conv_s2.conv_transpose_s2)(xg, wg, bg)
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    out.backward(g.to(DEV))
    rel = lambda a, r: float((a.detach().double().cpu() - r).abs().max() / r.abs().max())   # noqa: E731
    assert rel(out, ref.detach()) < 6e-3, rel(out, ref.detach())
    assert xg.grad.dtype == torch.bfloat16 and rel(xg.grad, xr.grad) < 8e-3, rel(xg.grad, xr.grad)
    low_w = ref.shape[3] if kind == "conv" else W      # width of the low-resolution tensor of the pair
    own = cin % 64 == 0 and cout % 64 == 0 and low_w % 4 == 0    # the kernel's weight gradient; else ATen's, in bf16
    assert wg.grad.dtype == torch.float32 and rel(wg.grad, wr.grad) < (3e-5 if own else 1e-2), rel(wg.grad, wr.grad)
    assert bg.grad.dtype == torch.float32 and rel(bg.grad, br.grad) < 1e-5


def test_eight_wave_convolution_kernels_still_agree_with_fp64():
    """The kernels the producer / consumer ones replaced stay in the library as the route for maps whose byte offsets do not
    fit the buffer-resource addressing (64 channels x H x W x 4 >= 2 GiB) and as the A/B partner (MMU_CONV3_WS=0,
    MMU_CONV_S2_WS=0 -- read once per process, hence a child process): 3 x 3, stride-2 and transposed convolution, forward
    and input gradient, ragged tiles, float32 and bfloat16, against float64."""
    import os, subprocess, sys
    code = r"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %r)
from mm_unet_amd import conv3x3_mfma, conv_s2
torch.manual_seed(7)
rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
x = torch.randn(2, 64, 40, 132, device="cuda", requires_grad=True)
w = (torch.randn(128, 64, 3, 3, device="cuda") / 24).requires_grad_()
b = torch.randn(128, device="cuda")
o = conv3x3_mfma.conv3x3_mfma(x, w, b); o.square().sum().backward()
xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
r = F.conv2d(xd, wd, b.double(), padding=1); r.square().sum().backward()
assert rel(o, r.detach()) < 1e-4 and rel(x.grad, xd.grad) < 1e-4, "3x3"
with torch.autocast("cuda", dtype=torch.bfloat16):
    ob = conv3x3_mfma.conv3x3_mfma(x.detach().to(torch.bfloat16), w.detach(), b)
assert ob.dtype == torch.bfloat16 and rel(ob, F.conv2d(x.detach().to(torch.bfloat16).double(), wd.detach(), b.double(), padding=1)) < 6e-3, "3x3 bf16"
for k in (3, 4):
    x = torch.randn(3, 64, 10, 130, device="cuda", requires_grad=True)
    w = (torch.randn(64, 64, k, k, device="cuda") / (8 * k)).requires_grad_()
    o = conv_s2.conv_s2(x, w, None); o.square().sum().backward()
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    r = F.conv2d(xd, wd, None, stride=2, padding=1); r.square().sum().backward()
    assert rel(o, r.detach()) < 1e-4 and rel(x.grad, xd.grad) < 1e-4, "stride 2"
x = torch.randn(2, 32, 9, 20, device="cuda", requires_grad=True)
w = (torch.randn(32, 64, 4, 4, device="cuda") / 24).requires_grad_()
o = conv_s2.conv_transpose_s2(x, w, None); o.square().sum().backward()
xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
r = F.conv_transpose2d(xd, wd, None, stride=2, padding=1); r.square().sum().backward()
assert rel(o, r.detach()) < 1e-4 and rel(x.grad, xd.grad) < 1e-4, "transposed"
print("eight-wave kernels ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MMU_CONV3_WS="0", MMU_CONV_S2_WS="0")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "eight-wave kernels ok" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


@pytest.mark.parametrize("case", [(2, 64, 64, 16, 32), (1, 64, 64, 64, 64), (2, 32, 64, 9, 20), (1, 128, 128, 6, 70)])
def test_conv_transpose_s2_mfma_vs_fp64(case):
    """csrc/conv_s2_mfma.hip, transposed form: ConvTranspose2d(4, stride 2, padding 1) (MMUNet.py:360) forward, input
    gradient (the strided kernel) and weight gradient against ATen in float64."""
    import torch.nn.functional as F
    from mm_unet_amd import conv_s2
    B, cin, cout, H, W = case
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(B, cin, H, W, generator=gen).to(DEV).requires_grad_()
    w = (torch.randn(cin, cout, 4, 4, generator=gen) / (4 * cin ** 0.5)).to(DEV).requires_grad_()
    b = torch.randn(cout, generator=gen).to(DEV).requires_grad_()
    assert conv_s2.convt_supported(x, w)
    out = conv_s2.conv_transpose_s2(x, w, b)
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    ref = F.conv_transpose2d(xd, wd, bd, stride=2, padding=1)
    assert out.shape == ref.shape
    g = torch.randn(out.shape, generator=gen).to(DEV)
    out.backward(g)
    ref.backward(g.double())
    sc = lambda t: max(1.0, float(t.abs().max()))   # noqa: E731
    assert float((out.double() - ref).abs().max()) < 1e-4 * sc(ref), "forward"
    assert float((x.grad.double() - xd.grad).abs().max()) < 1e-4 * sc(xd.grad), "d input"
    assert float((w.grad.double() - wd.grad).abs().max()) < 2e-4 * sc(wd.grad), "d weight"
    assert float((b.grad.double() - bd.grad).abs().max()) < 1e-4 * sc(bd.grad), "d bias"
