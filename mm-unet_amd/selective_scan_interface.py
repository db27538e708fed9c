"""Autograd layer over the HIP selective-scan / causal-conv1d kernels.

Public surface = that of the reference's ``mamba_ssm/ops/selective_scan_interface.py``
(names, positional order, defaults, return values):

    selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                      delta_softplus=False, return_last_state=False)        (:77-83)
    mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                   out_proj_weight, out_proj_bias, A, B=None, C=None, D=None, delta_bias=None,
                   B_proj_bias=None, C_proj_bias=None, delta_softplus=True)  (:606-614)
    mamba_inner_fn_no_out_proj(... same without out_proj ...)               (:627-633)
    bimamba_inner_fn(..., A, A_b, ...)                                      (:616-624)
    classes SelectiveScanFn, MambaInnerFn, MambaInnerFnNoOutProj, BiMambaInnerFn

What each fused function computes (forward, reference lines :159-224 / :296-365):
    x, z      = xz.chunk(2, dim=1)
    conv_out  = silu(causal_conv1d(x))                       -> HIP kernel
    x_dblT    = x_proj_weight @ conv_out                     (r+2N, B*L)  tokens-last, = reference x_dbl^T
    delta     = delta_proj_weight @ x_dblT[:r]               (D, B*L) viewed (B, D, L)
    B, C      = row blocks of x_dblT viewed (B, 1, N, L)     (strided views, no transpose copies)
    out_z     = selective_scan(conv_out, delta, A, B, C, D, z, delta_bias, softplus)  -> HIP kernels
    [out      = out_z^T @ out_proj_weight^T + out_proj_bias]
Backward recomputes conv_out and delta (the reference's checkpoint_lvl=1, :218-219,238-241; the small fused blocks
keep them instead) and
writes dx / dz straight into the two halves of one dxz buffer (:244-245).

No CPU path: tensors must live on the GPU, the HIP library must be present.
"""
import os

import torch
import torch.nn.functional as F

from . import _lib, causal_conv1d_hip, deferred, mfma_gemm, selective_scan_hip
from .tall_gemm import nt_splitk

try:  # torch >= 2.4
    from torch.amp import custom_bwd as _custom_bwd, custom_fwd as _custom_fwd

    def custom_fwd(fn):
        return _custom_fwd(fn, device_type="cuda")

    def custom_bwd(fn):
        return _custom_bwd(fn, device_type="cuda")
except ImportError:  # pragma: no cover
    from torch.cuda.amp import custom_bwd, custom_fwd


def _autocast_dtype():
    if torch.is_autocast_enabled():
        try:
            return torch.get_autocast_dtype("cuda")
        except AttributeError:  # pragma: no cover
            return torch.get_autocast_gpu_dtype()
    return None


def _unit_l(t):
    return t if t is None or t.stride(-1) == 1 else t.contiguous()


def _as_bnl4(t):
    """(batch, dstate, L) -> (batch, 1, dstate, L); returns (tensor, squeezed?)."""
    if t.dim() == 3:
        return t.unsqueeze(1), True
    return t, False


class SelectiveScanFn(torch.autograd.Function):
    """selective_scan_interface.py:14-74."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False):
        u, delta, B, C, z = _unit_l(u), _unit_l(delta), _unit_l(B), _unit_l(C), _unit_l(z)
        if D is not None:
            D = D.contiguous()
        B, ctx.squeeze_B = _as_bnl4(B)
        C, ctx.squeeze_C = _as_bnl4(C)
        out, x, *rest = selective_scan_hip.fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus)
        ctx.delta_softplus = delta_softplus
        ctx.has_z = z is not None
        last_state = x[:, :, -1, 1::2]  # (batch, dim, dstate)
        if not ctx.has_z:
            ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, x)
            return out if not return_last_state else (out, last_state)
        ctx.save_for_backward(u, delta, A, B, C, D, z, delta_bias, x, out)
        out_z = rest[0]
        return out_z if not return_last_state else (out_z, last_state)

    @staticmethod
    def backward(ctx, dout, *args):
        if not ctx.has_z:
            u, delta, A, B, C, D, delta_bias, x = ctx.saved_tensors
            z = out = None
        else:
            u, delta, A, B, C, D, z, delta_bias, x, out = ctx.saved_tensors
        dout = _unit_l(dout)
        du, ddelta, dA, dB, dC, dD, ddelta_bias, *rest = selective_scan_hip.bwd(
            u, delta, A, B, C, D, z, delta_bias, dout, x, out, None, ctx.delta_softplus, False)
        dz = rest[0] if ctx.has_z else None
        dB = dB.squeeze(1) if ctx.squeeze_B else dB
        dC = dC.squeeze(1) if ctx.squeeze_C else dC
        return (du, ddelta, dA, dB, dC, dD if D is not None else None, dz,
                ddelta_bias if delta_bias is not None else None, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """If return_last_state is True returns (out, last_state); last_state is (batch, dim, dstate) and
    carries no gradient (selective_scan_interface.py:77-83)."""
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)


# ---------------------------------------------------------------------------------------------
# fused inner function
# ---------------------------------------------------------------------------------------------
# conv1d + x_proj + dt_proj of the small (inner width 2 / 6, dt_rank 1) blocks as one kernel; False keeps the
# three-launch path (tests compare the two)
KEEP_LARGE_ACTIVATIONS = os.environ.get("MMUNET_KEEP_ACTIVATIONS", "1") != "0"
KEEP_MAX_BYTES = 1 << 30     # per tensor
# the scan's un-gated output kept for the backward kernel (what the reference saves); "0": recomputed there
KEEP_SCAN_OUT = os.environ.get("MMUNET_KEEP_SCAN_OUT", "1") != "0"
PRE_SMALL_FUSED = True
POST_SMALL_FUSED = True   # ... and the backward mirror (d x_dbl row 0, both weight gradients, d conv += W_x^T d x_dbl)


def _dbl_view(t):
    """(B, D, L) -> (D, B*L) matrix (a view when t is laid out [D][B][L], as inside mamba_inner)."""
    b, d, l = t.shape
    return t.permute(1, 0, 2).reshape(d, b * l)


def _rows_as_bnl(rows, batch, L):
    """(N, B*L) rows of a tokens-last matrix -> (B, 1, N, L) view (strides (L, *, B*L, 1)); the scan
    kernels take B/C with any batch/state stride, so no transpose copy is needed."""
    n = rows.shape[0]
    return rows.view(n, batch, L).permute(1, 0, 2).unsqueeze(1)


def _pre_small_ok(x, conv1d_weight, x_proj_weight, delta_proj_weight, B, C, B_proj_bias, C_proj_bias):
    """The one-kernel conv1d + x_proj + dt_proj (csrc/mamba_pre.hip) covers MMConv's Mamba blocks: inner width
    2 or 6, conv width 4, dt_rank 1, float32 everywhere, input-dependent B and C without projection biases."""
    return (x.dtype == torch.float32 and x_proj_weight.dtype == torch.float32
            and delta_proj_weight.dtype == torch.float32 and x.shape[1] in (2, 6) and conv1d_weight.shape[-1] == 4
            and delta_proj_weight.shape[1] == 1 and B is None and C is None and B_proj_bias is None
            and C_proj_bias is None and x.shape[2] % 4 == 0 and x.stride(2) == 1
            and x.stride(0) % 4 == 0 and x.stride(1) % 4 == 0 and x.data_ptr() % 16 == 0
            and conv1d_weight.dtype == torch.float32)


def _pre_small(x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, want_x_dbl):
    """Returns (conv1d_out, x_dblT or None, delta), conv1d_out/delta laid out [D][B][L]."""
    batch, dim, L = x.shape
    rows = x_proj_weight.shape[0]
    conv = torch.empty((dim, batch, L), dtype=torch.float32, device=x.device).permute(1, 0, 2)
    delta = torch.empty((dim, batch, L), dtype=torch.float32, device=x.device).permute(1, 0, 2)
    x_dblT = torch.empty((rows, batch * L), dtype=torch.float32, device=x.device) if want_x_dbl else None
    cw = conv1d_weight.contiguous()
    wx = x_proj_weight.contiguous()
    wdt = delta_proj_weight.contiguous()
    p = _lib.MambaPreParams()
    p.batch, p.dim, p.seqlen, p.rows = batch, dim, L, rows
    p.x, p.x_bs, p.x_ds = x.data_ptr(), x.stride(0), x.stride(1)
    p.conv_weight, p.conv_bias = cw.data_ptr(), _lib.ptr(conv1d_bias)
    p.x_proj_weight, p.dt_proj_weight = wx.data_ptr(), wdt.data_ptr()
    p.conv_out, p.conv_bs, p.conv_ds = conv.data_ptr(), conv.stride(0), conv.stride(1)
    p.x_dbl = _lib.ptr(x_dblT)
    p.delta, p.delta_bs, p.delta_ds = delta.data_ptr(), delta.stride(0), delta.stride(1)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mmu_mamba_pre_small(p, _lib.stream_of(x)))
    return conv, x_dblT, delta


def _mat_of(t, batch, L):
    """[D][B][L]-laid-out (B, D, L) tensor -> True when it is also a plain [D][B*L] matrix."""
    return t.stride(2) == 1 and t.stride(0) == L and t.stride(1) == batch * L and t.data_ptr() % 16 == 0


def _post_small(ddelta, x_dblT, dx_dblT, conv1d_out, dconv1d_out, x_proj_weight, delta_proj_weight):
    """csrc/mamba_pre.hip (backward mirror): returns (dx_proj_weight, ddelta_proj_weight); dconv1d_out is
    updated in place with W_x^T d x_dbl."""
    batch, dim, L = ddelta.shape
    rows = x_proj_weight.shape[0]
    tokens = batch * L
    dwx = torch.empty_like(x_proj_weight, memory_format=torch.contiguous_format)
    dwdt = torch.empty((dim, 1), dtype=torch.float32, device=ddelta.device)
    nws = _lib.lib().mmu_mamba_post_small_workspace_floats(dim, rows, tokens)
    ws = torch.empty(nws, dtype=torch.float32, device=ddelta.device)
    wx = x_proj_weight.contiguous()
    wdt = delta_proj_weight.contiguous()
    p = _lib.MambaPostParams()
    p.dim, p.rows, p.tokens = dim, rows, tokens
    p.ddelta, p.dt, p.dx_dbl = ddelta.data_ptr(), x_dblT.data_ptr(), dx_dblT.data_ptr()
    p.conv_out, p.dconv_out = conv1d_out.data_ptr(), dconv1d_out.data_ptr()
    p.x_proj_weight, p.dt_proj_weight = wx.data_ptr(), wdt.data_ptr()
    p.dx_proj_weight, p.ddt_proj_weight, p.workspace = dwx.data_ptr(), dwdt.data_ptr(), ws.data_ptr()
    with torch.cuda.device(ddelta.device), deferred.guard(deferred.may_defer(x_proj_weight, delta_proj_weight)):
        _lib.check(_lib.lib().mmu_mamba_post_small(p, _lib.stream_of(ddelta)))
    deferred.keep(ws)    # (inside a deferred.Scope the sum over the workgroups' partials runs later)
    return dwx, dwdt


OWN_PROJ = os.environ.get("MMUNET_OWN_PROJ", "1") != "0"   # "0": x_proj / dt_proj and their input gradients as library GEMMs


def _own_proj_ok(conv_m, x_proj_weight, delta_proj_weight, tokens):
    """x_proj / dt_proj (and their input gradients) on csrc/gemm_tokens_mfma.hip + csrc/dt_proj.hip: float32, a
    tokens-last conv matrix with unit token stride, dt_rank <= 8, enough tokens to give every CU a 512-token tile."""
    return (OWN_PROJ and tokens % 4 == 0 and tokens >= 16384 and conv_m.dim() == 2 and conv_m.stride(1) == 1
            and conv_m.stride(0) % 4 == 0 and x_proj_weight.is_contiguous() and delta_proj_weight.is_contiguous()
            and 1 <= delta_proj_weight.shape[1] <= 8 and x_proj_weight.shape[0] in mfma_gemm.X_PROJ_ROWS
            and x_proj_weight.shape[1] % 4 == 0 and x_proj_weight.shape[1] * 4 * ((x_proj_weight.shape[0] + 3) // 4 * 4) <= 65536
            and mfma_gemm.tokens_supported(conv_m, x_proj_weight, delta_proj_weight))


# bf16 activations (autocast): 0 = x_proj / dt_proj and their input gradients as library GEMMs on bf16 casts of the
# weights; 1 = x_proj on gemm_tokens' bf16 form, the rest on csrc/dt_proj.hip's streaming kernels (bf16 rows, float32
# weights and arithmetic); 2 = x_proj on the streaming kernel too
LOWP_PROJ = int(os.environ.get("MMUNET_LOWP_PROJ", "1"))


def _lowp_proj_ok(conv_m, x_proj_weight, delta_proj_weight):
    """x_proj / dt_proj (and their input gradients) of bf16 activations on this library's kernels: float32 contiguous
    weights, a tokens-last bf16 conv matrix with unit token stride.  (With all four products on gemm_tokens' 512-token
    kernel this lost to the library -- 59.76 against 58.46 ms per step of config 3: the products with 4 rows / inner 4
    pad to its tiles; they are streaming kernels now.)"""
    return (LOWP_PROJ > 0 and conv_m.dtype == torch.bfloat16 and conv_m.dim() == 2 and conv_m.shape[1] % 4 == 0
            and conv_m.stride(0) % 4 == 0 and conv_m.data_ptr() % 8 == 0
            and x_proj_weight.is_contiguous() and delta_proj_weight.is_contiguous() and 1 <= delta_proj_weight.shape[1] <= 8
            and delta_proj_weight.dtype == torch.float32 and mfma_gemm.tokens_lowp_supported(x_proj_weight, conv_m)
            and x_proj_weight.shape[0] in mfma_gemm.X_PROJ_ROWS and x_proj_weight.shape[1] % 4 == 0
            and x_proj_weight.shape[1] * 4 * ((x_proj_weight.shape[0] + 3) // 4 * 4) <= 65536)


def _project(conv1d_out, x_proj_weight, delta_proj_weight, d_state, B, C, B_proj_bias, C_proj_bias):
    """delta, B, C from the conv output (selective_scan_interface.py:181-210), computed tokens-last:
    x_dblT = W_x @ conv (r+2N, B*L) -- the transpose of the reference's x_dbl -- so that delta, B and C
    are row blocks of it and nothing has to be transposed."""
    batch, dim, L = conv1d_out.shape
    r = delta_proj_weight.shape[1]
    conv_m = _dbl_view(conv1d_out)
    if _own_proj_ok(conv_m, x_proj_weight, delta_proj_weight, batch * L):
        # W_x (36 x 128) as a zero-padded matrix-core product with masked rows (csrc/gemm_tokens_mfma.hip: 98 us at the
        # largest block; the streaming float32 kernel of csrc/dt_proj.hip takes 114, the library 97), W_dt (128 x 4) as
        # a streaming kernel (48 us; the library 76)
        T = batch * L
        x_dblT = torch.empty((x_proj_weight.shape[0], T), device=conv_m.device, dtype=torch.float32)
        mfma_gemm.gemm_tokens(x_proj_weight, conv_m, x_dblT, x_proj_weight.shape[0], dim, T, 1, conv_m.stride(0), 0, T, 0)
        delta = mfma_gemm.dt_proj(delta_proj_weight, x_dblT[:r]).view(dim, batch, L).permute(1, 0, 2)
    elif _lowp_proj_ok(conv_m, x_proj_weight, delta_proj_weight):
        # bf16 activations (autocast), float32 weights: x_proj on gemm_tokens' bf16 form, dt_proj (inner dimension 4) as
        # a streaming kernel -- one launch each instead of a cast of the weight + a library GEMM; products exact to the
        # weight's 16 / 24 bits instead of its 8
        T = batch * L
        rows = x_proj_weight.shape[0]
        x_dblT = torch.empty((rows, T), device=conv_m.device, dtype=torch.bfloat16)
        if LOWP_PROJ == 2:
            mfma_gemm.x_proj(x_proj_weight, conv_m, x_dblT)
        else:
            mfma_gemm.gemm_tokens(x_proj_weight, conv_m, x_dblT, rows, dim, T, 1, conv_m.stride(0), 0, T, 0)
        delta = mfma_gemm.dt_proj(delta_proj_weight, x_dblT[:r]).view(dim, batch, L).permute(1, 0, 2)
    else:
        if x_proj_weight.dtype != conv_m.dtype:      # (a caller that kept its float32 weights for the branch above)
            x_proj_weight, delta_proj_weight = x_proj_weight.to(conv_m.dtype), delta_proj_weight.to(conv_m.dtype)
        x_dblT = x_proj_weight @ conv_m                                  # (r + 2N, B*L)
        delta = (delta_proj_weight @ x_dblT[:r]).view(dim, batch, L).permute(1, 0, 2)
    if B is None:
        B = _rows_as_bnl(x_dblT[r:r + d_state], batch, L)
        if B_proj_bias is not None:
            B = B + B_proj_bias.to(dtype=B.dtype).view(1, 1, -1, 1)
    else:
        B = _unit_l(B)
    if C is None:
        C = _rows_as_bnl(x_dblT[r + d_state:], batch, L)
        if C_proj_bias is not None:
            C = C + C_proj_bias.to(dtype=C.dtype).view(1, 1, -1, 1)
    else:
        C = _unit_l(C)
    return x_dblT, delta, B, C


def _project_backward(ddelta, x_dblT, dx_dblT, conv1d_out, dconv1d_out, x_proj_weight, delta_proj_weight):
    """Backward of :func:`_project` once the scan has left d delta and (in rows r.. of ``dx_dblT``) dB / dC
    (selective_scan_interface.py:273-277): returns (d x_proj_weight, d delta_proj_weight, d conv1d_out), the last one
    = ``dconv1d_out + W_x^T d x_dbl`` (in place when ``dconv1d_out`` is laid out [D][B][L])."""
    batch, dim, L = ddelta.shape
    r = delta_proj_weight.shape[1]
    direct = dx_dblT.dtype == torch.float32
    ddelta_m = _dbl_view(ddelta)                                            # (D, B*L)
    ddelta_proj_weight = nt_splitk(ddelta_m, x_dblT[:r]).to(delta_proj_weight.dtype)  # (D, r)      (:273)
    conv_m = _dbl_view(conv1d_out)
    dconv_m = _dbl_view(dconv1d_out)                                        # (D, B*L)
    in_place = dconv_m.data_ptr() == dconv1d_out.data_ptr()
    own = (in_place and direct and dx_dblT.is_contiguous()
           and _own_proj_ok(conv_m, x_proj_weight, delta_proj_weight, batch * L)
           and mfma_gemm.dt_proj_supported(delta_proj_weight, dx_dblT[:r], ddelta_m)
           and mfma_gemm.tokens_supported(dconv_m, dx_dblT) and dconv_m.stride(1) == 1 and dconv_m.stride(0) % 4 == 0)
    lowp = (not own and in_place and dx_dblT.is_contiguous() and _lowp_proj_ok(conv_m, x_proj_weight, delta_proj_weight)
            and mfma_gemm.tokens_lowp_supported(x_proj_weight, ddelta_m, dconv_m, dx_dblT)
            and mfma_gemm.dt_proj_supported(delta_proj_weight, dx_dblT[:r], ddelta_m)
            and mfma_gemm.x_proj_supported(x_proj_weight, dconv_m, dx_dblT))
    T = batch * L
    if own:
        mfma_gemm.dt_proj_input_grad(delta_proj_weight, ddelta_m, dx_dblT[:r])  # (r, B*L)         (:274)
    elif lowp:
        mfma_gemm.dt_proj_input_grad(delta_proj_weight, ddelta_m, dx_dblT[:r])
    else:
        if delta_proj_weight.dtype != ddelta_m.dtype:
            x_proj_weight, delta_proj_weight = x_proj_weight.to(ddelta_m.dtype), delta_proj_weight.to(ddelta_m.dtype)
        torch.matmul(delta_proj_weight.t(), ddelta_m, out=dx_dblT[:r])     # (r, B*L)              (:274)
    dx_proj_weight = nt_splitk(dx_dblT, conv_m).to(x_proj_weight.dtype)    # (r+2N, D)             (:276)
    if own:
        mfma_gemm.x_proj_input_grad_add(x_proj_weight, dx_dblT, dconv_m)   # d conv += W_x^T d x_dbl, in place   (:277)
    elif lowp:
        mfma_gemm.x_proj_input_grad_add(x_proj_weight, dx_dblT, dconv_m)   # (bf16 rows, float32 sums, one rounding)
    elif in_place:
        dconv_m.addmm_(x_proj_weight.t(), dx_dblT)                          # in place              (:277)
    else:  # dconv1d_out was not [D][B][L]; keep it correct anyway
        dconv_m = torch.addmm(dconv_m, x_proj_weight.t(), dx_dblT)
    return dx_proj_weight, ddelta_proj_weight, dconv_m.view(dim, batch, L).permute(1, 0, 2)


def _inner_forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                   out_proj_bias, A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, checkpoint_lvl,
                   with_out_proj):
    assert checkpoint_lvl in (0, 1)
    if A.is_complex():
        raise RuntimeError("mamba_inner_fn: complex A is not supported (not on the MM-UNet path)")
    d_state = A.shape[-1]
    ac = _autocast_dtype()
    if ac is not None:  # :169-171, :307-312
        x_proj_weight = x_proj_weight.to(dtype=ac)
        delta_proj_weight = delta_proj_weight.to(dtype=ac)
        if with_out_proj:
            out_proj_weight = out_proj_weight.to(dtype=ac)
            out_proj_bias = out_proj_bias.to(dtype=ac) if out_proj_bias is not None else None
    xz = _unit_l(xz)
    conv1d_weight = conv1d_weight.view(conv1d_weight.shape[0], conv1d_weight.shape[-1])  # "d 1 w -> d w"
    x, z = xz.chunk(2, dim=1)
    conv1d_bias = conv1d_bias.contiguous() if conv1d_bias is not None else None
    ctx.is_variable_B = B is None
    ctx.is_variable_C = C is None
    # A = -exp(A_log) from mamba_simple's batched form: the scan returns d A_log = dA * A itself (one multiply launch per
    # model less, and the sum may then wait for deferred.Scope.launch() with the other parameter gradients -- which it
    # may only if D and delta_bias are parameters too, i.e. nothing reads their gradients during the backward pass)
    ctx.A_neg_exp = bool(getattr(A, "_mmu_neg_exp", False)) and A.dtype == torch.float32 and A.is_contiguous()
    ctx.scan_params_are_leaves = all(t is None or (t.is_leaf and t.dtype == torch.float32) for t in (D, delta_bias))
    ctx.B_proj_bias_is_None = B_proj_bias is None
    ctx.C_proj_bias_is_None = C_proj_bias is None
    _lib.require_gpu(x)
    ctx.pre_small = (PRE_SMALL_FUSED and (conv1d_bias is None or conv1d_bias.dtype == torch.float32)
                     and _pre_small_ok(x, conv1d_weight, x_proj_weight, delta_proj_weight, B, C, B_proj_bias,
                                       C_proj_bias))
    if ctx.pre_small:
        batch_, _, L_ = x.shape
        conv1d_out, x_dblT, delta = _pre_small(x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, True)
        B = _rows_as_bnl(x_dblT[1:1 + d_state], batch_, L_)
        C = _rows_as_bnl(x_dblT[1 + d_state:], batch_, L_)
    else:
        conv1d_out = causal_conv1d_hip.causal_conv1d_fwd(x, conv1d_weight, conv1d_bias, True)
        x_dblT, delta, B, C = _project(conv1d_out, x_proj_weight, delta_proj_weight, d_state, B, C, B_proj_bias,
                                       C_proj_bias)
    if D is not None:
        D = D.contiguous()
    # The un-gated `out` (y + D u) is kept for the backward where the 512-token backward kernel will read it (d_state 16,
    # L a multiple of 512: selective_scan.cpp:338 `out_`, as the reference does) -- that kernel then neither recomputes
    # C h per state nor sums it over its eight waves.  Elsewhere the backward recomputes y from the states it rebuilds.
    keep_out = (KEEP_SCAN_OUT and d_state == 16 and conv1d_out.shape[-1] % 512 == 0
                and conv1d_out.numel() * conv1d_out.element_size() <= KEEP_MAX_BYTES)
    scan_out, scan_intermediates, out_z = selective_scan_hip.fwd(conv1d_out, delta, A, B, C, D, z, delta_bias,
                                                                 delta_softplus, want_out=keep_out, opaque_x=True)
    ctx.delta_softplus = delta_softplus
    ctx.checkpoint_lvl = checkpoint_lvl
    ctx.with_out_proj = with_out_proj
    ctx.out_proj_bias_is_None = out_proj_bias is None
    # The reference drops conv1d_out and delta and recomputes them in the backward pass (checkpoint_lvl=1, :218-219,238-241:
    # a memory saving sized for 24-80 GB devices).  One MI355X has 288 GB: at BASELINE's sizes the nine large scans of a
    # step keep 2 x 268 MB each at most (2.1 GB in all), and the recomputation is a conv1d kernel + a dt_proj GEMM per
    # scan and step (0.8 ms of 40).  KEEP_LARGE_ACTIVATIONS = False restores the reference's behaviour.
    if checkpoint_lvl >= 1 and not ctx.pre_small and not (KEEP_LARGE_ACTIVATIONS and
                                                          conv1d_out.numel() * conv1d_out.element_size() <= KEEP_MAX_BYTES):
        conv1d_out, delta = None, None
    # (the small blocks keep both: 2 x 6 channels per token against one more launch per block and step -- the launch
    #  chain, not memory, is what the 47 small Mamba blocks cost)
    # B and C are views of x_dblT when they are input-dependent: save the matrix, rebuild the views
    ctx.save_for_backward(xz, conv1d_weight, conv1d_bias, x_dblT, x_proj_weight, delta_proj_weight,
                          out_proj_weight if with_out_proj else None, conv1d_out, delta, A,
                          None if ctx.is_variable_B else B, None if ctx.is_variable_C else C, D, delta_bias,
                          scan_intermediates, scan_out if keep_out else None)
    if not with_out_proj:
        return out_z
    # (B, L, E) = out_z^T W_out^T, computed tokens-last then viewed
    y = out_proj_weight @ _dbl_view(out_z)                                # (E, B*L)
    if out_proj_bias is not None:
        y = y + out_proj_bias.view(-1, 1)
    batch, _, L = out_z.shape
    return y.view(-1, batch, L).permute(1, 2, 0)


def _inner_backward(ctx, dout):
    (xz, conv1d_weight, conv1d_bias, x_dblT, x_proj_weight, delta_proj_weight, out_proj_weight, conv1d_out, delta,
     A, B, C, D, delta_bias, scan_intermediates, scan_out) = ctx.saved_tensors
    batch, _, L = xz.shape
    r = delta_proj_weight.shape[1]
    d_state = A.shape[-1]
    x, z = xz.chunk(2, dim=1)
    dim = x.shape[1]
    if ctx.checkpoint_lvl == 1 and ctx.pre_small:
        if conv1d_out is None:
            conv1d_out, _, delta = _pre_small(x, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, False)
    elif ctx.checkpoint_lvl == 1 and conv1d_out is None:
        conv1d_out = causal_conv1d_hip.causal_conv1d_fwd(x, conv1d_weight, conv1d_bias, True)
        delta = (delta_proj_weight @ x_dblT[:r]).view(dim, batch, L).permute(1, 0, 2)
    if ctx.is_variable_B:
        B = _rows_as_bnl(x_dblT[r:r + d_state], batch, L)
    if ctx.is_variable_C:
        C = _rows_as_bnl(x_dblT[r + d_state:], batch, L)
    dxz = torch.empty_like(xz)
    dx, dz = dxz.chunk(2, dim=1)
    if ctx.with_out_proj:
        # dout: (B, L, E) -> (E, B*L); dy = W_out^T dout laid out [D][B][L]   (:387-388)
        dout_m = dout.permute(2, 0, 1).reshape(dout.shape[-1], batch * L)
        dout_y = (out_proj_weight.t() @ dout_m).view(dim, batch, L).permute(1, 0, 2)
    else:
        dout_m = None
        dout_y = _unit_l(dout)
    # projection-gradient matrix; fp32 dB/dC are written straight into its rows by the scan kernel
    dx_dblT = torch.empty_like(x_dblT)
    direct = dx_dblT.dtype == torch.float32
    dB_out = _rows_as_bnl(dx_dblT[r:r + d_state], batch, L) if (ctx.is_variable_B and direct) else None
    dC_out = _rows_as_bnl(dx_dblT[r + d_state:], batch, L) if (ctx.is_variable_C and direct) else None
    res = selective_scan_hip.bwd(conv1d_out, delta, A, B, C, D, z, delta_bias, dout_y, scan_intermediates, scan_out, dz,
                                 ctx.delta_softplus, ctx.with_out_proj, dB_out=dB_out, dC_out=dC_out,
                                 dA_times_A=ctx.A_neg_exp, defer=ctx.A_neg_exp and ctx.scan_params_are_leaves)
    dconv1d_out, ddelta, dA, dB, dC, dD, ddelta_bias, dz = res[:8]
    if ctx.A_neg_exp:
        deferred.mark_prescaled(dA)
    dout_proj_weight = dout_proj_bias = None
    if ctx.with_out_proj:
        out_z = res[8]
        dout_proj_weight = nt_splitk(dout_m, _dbl_view(out_z)).to(out_proj_weight.dtype)  # (E, D)   (:394)
        dout_proj_bias = dout.sum(dim=(0, 1)) if not ctx.out_proj_bias_is_None else None
    dB_proj_bias = dC_proj_bias = None
    if ctx.is_variable_B:
        if dB_out is None:
            _rows_as_bnl(dx_dblT[r:r + d_state], batch, L).copy_(dB)
        dB_proj_bias = dx_dblT[r:r + d_state].sum(1) if not ctx.B_proj_bias_is_None else None
        dB = None
    if ctx.is_variable_C:
        if dC_out is None:
            _rows_as_bnl(dx_dblT[r + d_state:], batch, L).copy_(dC)
        dC_proj_bias = dx_dblT[r + d_state:].sum(1) if not ctx.C_proj_bias_is_None else None
        dC = None
    post_small = (ctx.pre_small and POST_SMALL_FUSED and direct and x_dblT.shape[0] == 33 and x_dblT.is_contiguous()
                  and dx_dblT.is_contiguous() and _mat_of(ddelta, batch, L) and _mat_of(conv1d_out, batch, L)
                  and _mat_of(dconv1d_out, batch, L) and ddelta.dtype == torch.float32
                  and dconv1d_out.dtype == torch.float32)
    if post_small:
        dx_proj_weight, ddelta_proj_weight = _post_small(ddelta, x_dblT, dx_dblT, conv1d_out, dconv1d_out,
                                                         x_proj_weight, delta_proj_weight)
    else:
        dx_proj_weight, ddelta_proj_weight, dconv1d_out = _project_backward(
            ddelta, x_dblT, dx_dblT, conv1d_out, dconv1d_out, x_proj_weight, delta_proj_weight)
    dx, dconv1d_weight, dconv1d_bias = causal_conv1d_hip.causal_conv1d_bwd(x, conv1d_weight, conv1d_bias,
                                                                           dconv1d_out, dx, True)
    dconv1d_bias = dconv1d_bias if conv1d_bias is not None else None
    dconv1d_weight = dconv1d_weight.unsqueeze(1)  # "d w -> d 1 w"
    return dict(dxz=dxz, dconv1d_weight=dconv1d_weight, dconv1d_bias=dconv1d_bias, dx_proj_weight=dx_proj_weight,
                ddelta_proj_weight=ddelta_proj_weight, dout_proj_weight=dout_proj_weight,
                dout_proj_bias=dout_proj_bias, dA=dA, dB=dB, dC=dC, dD=dD if D is not None else None,
                ddelta_bias=ddelta_bias if delta_bias is not None else None, dB_proj_bias=dB_proj_bias,
                dC_proj_bias=dC_proj_bias)


class MambaInnerFnNoOutProj(torch.autograd.Function):
    """selective_scan_interface.py:155-289 -- returns out_z (batch, dim, L)."""

    @staticmethod
    @custom_fwd
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None, D=None,
                delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True, checkpoint_lvl=1):
        return _inner_forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, None, None, A,
                              B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, checkpoint_lvl, False)

    @staticmethod
    @custom_bwd
    def backward(ctx, dout):
        g = _inner_backward(ctx, dout)
        return (g["dxz"], g["dconv1d_weight"], g["dconv1d_bias"], g["dx_proj_weight"], g["ddelta_proj_weight"],
                g["dA"], g["dB"], g["dC"], g["dD"], g["ddelta_bias"], g["dB_proj_bias"], g["dC_proj_bias"], None,
                None)


class MambaInnerFn(torch.autograd.Function):
    """selective_scan_interface.py:292-434 -- returns out_proj(out_z) (batch, L, d_model)."""

    @staticmethod
    @custom_fwd
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                out_proj_bias, A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                delta_softplus=True, checkpoint_lvl=1):
        return _inner_forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                              out_proj_weight, out_proj_bias, A, B, C, D, delta_bias, B_proj_bias, C_proj_bias,
                              delta_softplus, checkpoint_lvl, True)

    @staticmethod
    @custom_bwd
    def backward(ctx, dout):
        g = _inner_backward(ctx, dout)
        return (g["dxz"], g["dconv1d_weight"], g["dconv1d_bias"], g["dx_proj_weight"], g["ddelta_proj_weight"],
                g["dout_proj_weight"], g["dout_proj_bias"], g["dA"], g["dB"], g["dC"], g["dD"], g["ddelta_bias"],
                g["dB_proj_bias"], g["dC_proj_bias"], None, None)


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                   out_proj_bias, A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                   delta_softplus=True):
    return MambaInnerFn.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                              out_proj_bias, A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus)


def mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None,
                               D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    return MambaInnerFnNoOutProj.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C,
                                       D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus)


def bimamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                     out_proj_bias, A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                     C_proj_bias=None, delta_softplus=True):
    """Shared-projection bidirectional variant (selective_scan_interface.py:437-603; exported by the
    reference, not called by MM-UNet).  Built from the differentiable primitives following
    bimamba_inner_ref (:673-709): one conv + projection, a forward scan with A and a scan over the
    flipped sequence with A_b, summed, then out_proj."""
    from .causal_conv1d_interface import causal_conv1d_fn
    ac = _autocast_dtype()
    if ac is not None:
        x_proj_weight, delta_proj_weight = x_proj_weight.to(ac), delta_proj_weight.to(ac)
        out_proj_weight = out_proj_weight.to(ac)
        out_proj_bias = out_proj_bias.to(ac) if out_proj_bias is not None else None
    d_state = A.shape[-1]
    x, z = _unit_l(xz).chunk(2, dim=1)
    x = causal_conv1d_fn(x, conv1d_weight.view(conv1d_weight.shape[0], -1), conv1d_bias, "silu")
    _, delta, B, C = _project(x, x_proj_weight, delta_proj_weight, d_state, B, C, B_proj_bias, C_proj_bias)
    y = selective_scan_fn(x, delta, A, B, C, D, z=z, delta_bias=delta_bias, delta_softplus=delta_softplus)
    y_b = selective_scan_fn(x.flip([-1]), delta.flip([-1]), A_b, B.flip([-1]), C.flip([-1]), D, z.flip([-1]),
                            delta_bias, delta_softplus=delta_softplus)
    y = y + y_b.flip([-1])
    return F.linear(y.permute(0, 2, 1), out_proj_weight, out_proj_bias)


class BiMambaInnerFn:
    """Name kept for API parity (selective_scan_interface.py:437); ``apply`` forwards to
    :func:`bimamba_inner_fn` (autograd comes from the primitives it is built of)."""

    @staticmethod
    def apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias, A,
              A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
              delta_softplus=True, checkpoint_lvl=1):
        return bimamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                                out_proj_bias, A, A_b, B, C, D, delta_bias, B_proj_bias, C_proj_bias,
                                delta_softplus)
