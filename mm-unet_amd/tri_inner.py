"""The three ``mamba_inner`` calls of the tri-directional ("v3") Mamba block as ONE autograd function (f2).

requirements/mamba_simple.py:212-270::

    out   = mamba_inner_fn_no_out_proj(xz, conv1d, x_proj, dt_proj, A, ..., D, dt_bias)
    out_b = mamba_inner_fn_no_out_proj(xz.flip(-1), conv1d_b, ...)
    out_s = mamba_inner_fn_no_out_proj(slice_interleave(xz), conv1d_s, ...)
    total = out + out_b.flip(-1) + unslice(out_s)                    # -> out_proj

Here (csrc/tri_fused.hip): one kernel reads the x half of ``xz`` once and writes the three conv1d+SiLU outputs in their
scan orders; per direction x_proj / dt_proj / selective scan WITHOUT z (selective_scan_interface.py:181-215 with
``z = None``); one kernel forms ``silu(z) * (y + unflip(y_b) + unslice(y_s))`` -- the same value, since each direction's
gate is the same z in that direction's order.  The re-ordered copies of xz (mamba_simple.py:229,245-247) and the three
gated outputs never exist.  The backward mirrors it: d total -> dz + the three dy (one kernel), scan / projection
backward per direction, the three d conv_out -> dx + the conv weight gradients (one kernel).

Taken by ``Mamba._v3`` when the caller does not want the per-direction outputs o_1..o_3 (RCG, MMUNet.py:409), in
float32 outside autocast; everything else keeps the three-call route (tests compare the two).
"""
import torch

from . import _lib, deferred, selective_scan_hip
from .selective_scan_interface import _project, _project_backward, _rows_as_bnl

ENABLED = True   # False: Mamba._v3 keeps the three mamba_inner calls (fused_paths.plain_aten; tests)


def supported(xz, nslices, conv_weights, params):
    """float32 activations outside autocast, or bfloat16 activations (what the block's in_proj emits under bf16 autocast:
    selective_scan_interface.py:158,169-171 -- x_proj / dt_proj are then cast to bfloat16 as the reference casts them);
    float32 parameters, conv width 4, 4..64 slices dividing L, (batch, channel) rows of ``xz`` dense."""
    if not ENABLED or xz.dim() != 3 or not xz.is_cuda or xz.dtype not in (torch.float32, torch.bfloat16):
        return False
    if torch.is_autocast_enabled() and (torch.get_autocast_dtype("cuda") != torch.bfloat16 or xz.dtype != torch.bfloat16):
        return False
    B, C2, L = xz.shape
    if xz.stride(2) != 1 or C2 % 2 or not (4 <= nslices <= 64) or L % nslices or B * (C2 // 2) >= 65536:
        return False
    if any(w.shape[-1] != 4 or w.dtype != torch.float32 for w in conv_weights):
        return False
    return all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in params)


def _cbl(batch, dim, L, ref):
    """(batch, dim, L) tensor of ``ref``'s type laid out [dim][batch][L]: what the tokens-last projections want (a plain
    matrix view)."""
    return torch.empty((dim, batch, L), device=ref.device, dtype=ref.dtype).permute(1, 0, 2)


def _conv_params(x, ns, weights, biases):
    p = _lib.TriConvParams()
    p.batch, p.dim, p.seqlen, p.nslices = x.shape[0], x.shape[1], x.shape[2], ns
    p.dtype = _lib.dtype_code(x)
    p.x, p.x_bs, p.x_ds = x.data_ptr(), x.stride(0), x.stride(1)
    p.weight_f, p.weight_b, p.weight_s = (w.data_ptr() for w in weights)
    p.bias_f, p.bias_b, p.bias_s = (_lib.ptr(b) for b in biases)
    return p


def tri_conv_fwd(x, ns, weights, biases):
    """x (B, D, L), three (D, 4) weights / (D,) biases -> three silu(conv1d) outputs (B, D, L) laid out [D][B][L], in
    scan order: natural, flipped, slice-interleaved."""
    B, D, L = x.shape
    outs = [_cbl(B, D, L, x) for _ in range(3)]
    p = _conv_params(x, ns, weights, biases)
    p.out_f, p.out_b, p.out_s = (o.data_ptr() for o in outs)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mmu_tri_conv_fwd(p, _lib.stream_of(x)))
    return outs


def tri_conv_bwd(x, ns, weights, biases, douts, dx):
    """The three d conv_out ([D][B][L], scan order) -> dx (written into ``dx``, any batch / channel stride) and the
    three (dweight (D, 4), dbias (D,) or None)."""
    B, D, L = x.shape
    f32 = dict(device=x.device, dtype=torch.float32)
    dws = [torch.empty((D, 4), **f32) for _ in range(3)]
    dbs = [torch.empty((D,), **f32) if b is not None else None for b in biases]
    ws = torch.empty(_lib.lib().mmu_tri_conv_bwd_workspace_floats(B, D, L, ns), **f32)
    p = _conv_params(x, ns, weights, biases)
    p.dout_f, p.dout_b, p.dout_s = (g.data_ptr() for g in douts)
    p.dx, p.dx_bs, p.dx_ds = dx.data_ptr(), dx.stride(0), dx.stride(1)
    p.dweight_f, p.dweight_b, p.dweight_s = (t.data_ptr() for t in dws)
    p.dbias_f, p.dbias_b, p.dbias_s = (_lib.ptr(t) for t in dbs)
    p.workspace = ws.data_ptr()
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mmu_tri_conv_bwd(p, _lib.stream_of(x)))
    deferred.keep(ws)    # (inside a deferred.Scope the ordered sums of the block partials run later)
    return dws, dbs


def _gate_params(z, ns, ys):
    p = _lib.TriGateParams()
    p.batch, p.dim, p.seqlen, p.nslices = z.shape[0], z.shape[1], z.shape[2], ns
    p.dtype = _lib.dtype_code(z)
    p.z, p.z_bs, p.z_ds = z.data_ptr(), z.stride(0), z.stride(1)
    p.y_f, p.y_b, p.y_s = (y.data_ptr() for y in ys)
    return p


def _is_cbl(t):
    B, D, L = t.shape
    return t.stride(2) == 1 and t.stride(0) == L and t.stride(1) == B * L


def tri_gate_fwd(z, ns, ys):
    B, D, L = z.shape
    out = _cbl(B, D, L, z)
    p = _gate_params(z, ns, ys)
    p.out, p.out_bs, p.out_ds = out.data_ptr(), out.stride(0), out.stride(1)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().mmu_tri_gate_fwd(p, _lib.stream_of(z)))
    return out


def tri_gate_bwd(z, ns, ys, dout, dz):
    B, D, L = z.shape
    dys = [_cbl(B, D, L, z) for _ in range(3)]
    p = _gate_params(z, ns, ys)
    p.dout, p.dout_bs, p.dout_ds = dout.data_ptr(), dout.stride(0), dout.stride(1)
    p.dz, p.dz_bs, p.dz_ds = dz.data_ptr(), dz.stride(0), dz.stride(1)
    p.dy_f, p.dy_b, p.dy_s = (t.data_ptr() for t in dys)
    with torch.cuda.device(z.device):
        _lib.check(_lib.lib().mmu_tri_gate_bwd(p, _lib.stream_of(z)))
    return dys


N_PER_DIR = 7   # conv1d.weight, conv1d.bias, x_proj.weight, dt_proj.weight, A, D, dt_proj.bias


class TriMambaInnerFn(torch.autograd.Function):
    """forward(xz, nslices, *[conv_w, conv_b, x_proj_w, dt_proj_w, A, D, delta_bias] x (forward, flipped, sliced))
    -> total (B, d_inner, L) laid out [d_inner][B][L]."""

    @staticmethod
    def forward(ctx, xz, ns, *params):
        with torch.autocast("cuda", enabled=False):     # every cast below is explicit
            return TriMambaInnerFn._forward(ctx, xz, ns, *params)

    @staticmethod
    def _forward(ctx, xz, ns, *params):
        assert len(params) == 3 * N_PER_DIR
        dirs = [params[k * N_PER_DIR:(k + 1) * N_PER_DIR] for k in range(3)]
        ctx.proj_dtypes = [(d[2].dtype, d[3].dtype) for d in dirs]
        # (bf16 activations: _project takes the float32 x_proj / dt_proj weights on gemm_tokens' bf16 form, or casts them
        #  to bf16 for the library as selective_scan_interface.py:169-171 does)
        B, C2, L = xz.shape
        D_in = C2 // 2
        x, z = xz.chunk(2, dim=1)
        cws = [d[0].view(D_in, 4) if d[0].is_contiguous() else d[0].reshape(D_in, 4).contiguous() for d in dirs]
        cbs = [None if d[1] is None else d[1].contiguous() for d in dirs]
        convs = tri_conv_fwd(x, ns, cws, cbs)
        saved, ys = [], []
        ctx.flags = []
        for conv, d in zip(convs, dirs):
            _, _, xw, dtw, A, Dp, dbias = d
            d_state = A.shape[-1]
            x_dblT, delta, Bm, Cm = _project(conv, xw, dtw, d_state, None, None, None, None)
            Dc = Dp.contiguous() if Dp is not None else None
            y, inter = selective_scan_hip.fwd(conv, delta, A, Bm, Cm, Dc, None, dbias, True, want_out=True, opaque_x=True)[:2]
            ys.append(y)
            a_neg_exp = bool(getattr(A, "_mmu_neg_exp", False)) and A.dtype == torch.float32 and A.is_contiguous()
            leaves = all(t is None or (t.is_leaf and t.dtype == torch.float32) for t in (Dp, dbias))
            ctx.flags.append((a_neg_exp, leaves))
            saved += [conv, delta, x_dblT, inter, y, xw, dtw, A, Dc, dbias]
        ctx.ns = ns
        # the conv weight sums may wait for deferred.Scope.launch() only if autograd stores them straight into .grad
        ctx.conv_leaves = all(t is None or t.is_leaf for d in dirs for t in d[:2])
        ctx.save_for_backward(xz, *cws, *cbs, *saved)
        return tri_gate_fwd(z, ns, ys)

    @staticmethod
    def backward(ctx, dtotal):
        xz, *rest = ctx.saved_tensors
        cws, cbs, rest = rest[:3], rest[3:6], rest[6:]
        ns = ctx.ns
        B, C2, L = xz.shape
        D_in = C2 // 2
        x, z = xz.chunk(2, dim=1)
        per = [rest[k * 10:(k + 1) * 10] for k in range(3)]
        dxz = torch.empty_like(xz)
        dx, dz = dxz.chunk(2, dim=1)
        if dtotal.dtype != xz.dtype:
            dtotal = dtotal.to(xz.dtype)
        if dtotal.stride(2) != 1:
            dtotal = dtotal.contiguous()
        dys = tri_gate_bwd(z, ns, [p[4] for p in per], dtotal, dz)
        dconvs, grads = [], []
        for k, (conv, delta, x_dblT, inter, y, xw, dtw, A, Dc, dbias) in enumerate(per):
            a_neg_exp, leaves = ctx.flags[k]
            d_state = A.shape[-1]
            r = dtw.shape[1]
            Bm = _rows_as_bnl(x_dblT[r:r + d_state], B, L)
            Cm = _rows_as_bnl(x_dblT[r + d_state:], B, L)
            dx_dblT = torch.empty_like(x_dblT)
            direct = dx_dblT.dtype == torch.float32
            dB_out = _rows_as_bnl(dx_dblT[r:r + d_state], B, L) if direct else None
            dC_out = _rows_as_bnl(dx_dblT[r + d_state:], B, L) if direct else None
            res = selective_scan_hip.bwd(conv, delta, A, Bm, Cm, Dc, None, dbias, dys[k], inter, None, None, True, False,
                                         dB_out=dB_out, dC_out=dC_out, dA_times_A=a_neg_exp, defer=a_neg_exp and leaves)
            dconv, ddelta, dA, dBm, dCm, dD, ddbias = res[:7]
            if a_neg_exp:
                deferred.mark_prescaled(dA)
            if dB_out is None:
                _rows_as_bnl(dx_dblT[r:r + d_state], B, L).copy_(dBm)
                _rows_as_bnl(dx_dblT[r + d_state:], B, L).copy_(dCm)
            dxw, ddtw, dconv = _project_backward(ddelta, x_dblT, dx_dblT, conv, dconv, xw, dtw)
            if not (dconv.stride(2) == 1 and dconv.stride(0) == L and dconv.stride(1) == B * L):
                t = _cbl(B, D_in, L, dconv)
                t.copy_(dconv)
                dconv = t
            dconvs.append(dconv)
            grads.append([None, None, dxw.to(ctx.proj_dtypes[k][0]), ddtw.to(ctx.proj_dtypes[k][1]), dA,
                          dD if Dc is not None else None, ddbias if dbias is not None else None])
        if ctx.conv_leaves:
            dws, dbs = tri_conv_bwd(x, ns, cws, cbs, dconvs, dx)
        else:
            with deferred.paused():
                dws, dbs = tri_conv_bwd(x, ns, cws, cbs, dconvs, dx)
        out = [dxz, None]
        for k in range(3):
            grads[k][0] = dws[k].unsqueeze(1)        # "d w -> d 1 w"
            grads[k][1] = dbs[k]
            out += grads[k]
        return tuple(out)


def tri_mamba_inner(xz, nslices, dirs):
    """``dirs``: three tuples (conv1d_weight (D, 1, 4), conv1d_bias, x_proj_weight, dt_proj_weight, A, D, delta_bias) for
    the natural, flipped and slice-interleaved direction."""
    flat = [t for d in dirs for t in d]
    return TriMambaInnerFn.apply(xz, nslices, *flat)
