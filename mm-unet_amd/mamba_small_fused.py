"""The K-channel Mamba chain of an MMConv block on a small map as one kernel each way (csrc/mamba_small_fused.hip).

``mamba_rows(offset, mamba, altho, extend_scope)`` -> row coordinates ``(B, K, H, W)``: what
``MMConv.get_coordinate_map_2D(..., rows_only=True)`` returns (src/UM_Net/MMUNet.py:122-188) --

    y_off -> two-row zig-zag flatten (:68-93,178-180) -> ``self.mamba`` (uni-directional branch of
    requirements/mamba_simple.py:303-318 = ``MambaInnerFn``, selective_scan_interface.py:292-434) -> inverse zig-zag
    (:95-121,182-183) -> ``clamp(softplus(altho), .01) * y_keep + row + extend_scope * cumsum-from-centre`` (:138-188)

for power-of-two maps of 64 .. 1,024 tokens (MM-UNet's 16 x 16 and 32 x 32 maps), instead of the
six-launch chain zigzag_inproj -> mamba_pre_small -> selective scan (3 kernels) -> coords_outproj and its eleven-launch
backward.  float32; taps K in (1, 3); conv width 4; dt_rank 1; d_state <= 64; no in_proj / out_proj / x_proj bias.
Anything else: ``supported()`` is False and MMConv takes the chain.  No CPU path.
"""
import os

import torch

from . import _lib

ENABLED = os.environ.get("MMUNET_SMALL_FUSED", "1") != "0"   # False / "0": MMConv keeps the multi-kernel chain (A/B runs, tests)


def supported(offset, K, mamba):
    if not (ENABLED and offset.is_cuda and offset.dtype == torch.float32 and offset.dim() == 4 and K in (1, 3)
            and offset.shape[1] == 2 * K):
        return False
    m = mamba
    if not (m.d_conv == 4 and m.dt_rank == 1 and m.d_inner == 2 * K and m.in_proj.bias is None
            and m.out_proj.bias is None and m.x_proj.bias is None and m.dt_proj.bias is not None
            and all(t.dtype == torch.float32 for t in (m.in_proj.weight, m.conv1d.weight, m.x_proj.weight,
                                                       m.dt_proj.weight, m.out_proj.weight, m.A_log, m.D))):
        return False
    return bool(_lib.lib().mmu_mamba_small_supported(K, offset.shape[2], offset.shape[3], m.d_state))


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _need(t, shape, what):
    if t.dtype != torch.float32 or tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"mamba_small_fused: {what} must be float32 {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")
    return _c(t)


class MambaSmallFusedFn(torch.autograd.Function):
    """Inputs as the reference's modules hold them: in_proj.weight (4K, K), conv1d.weight (2K, 1, 4), conv1d.bias (2K) or
    None, x_proj.weight (1 + 2N, 2K), dt_proj.weight (2K, 1), dt_proj.bias (2K), A = -exp(A_log) (2K, N), D (2K),
    out_proj.weight (K, 2K), altho (scalar)."""

    @staticmethod
    def forward(ctx, offset, w_in, conv_w, conv_b, w_x, w_dt, dt_bias, A, D, w_out, altho, scope):
        _lib.require_gpu(offset, w_in, conv_w, conv_b, w_x, w_dt, dt_bias, A, D, w_out, altho)
        if offset.dim() != 4 or offset.dtype != torch.float32 or offset.shape[1] not in (2, 6):
            raise RuntimeError(f"mamba_small_fused: offset must be float32 (B, 2K, H, W) with K in (1, 3), got "
                               f"{offset.dtype} {tuple(offset.shape)}")
        B, C2, H, W = offset.shape
        K = C2 // 2
        Dn = 2 * K
        if A.dim() != 2:
            raise RuntimeError("mamba_small_fused: A must be (2K, N)")
        N = A.shape[1]
        L = _lib.lib()
        if not L.mmu_mamba_small_supported(K, H, W, N):
            raise RuntimeError(f"mamba_small_fused: unsupported shape (K={K}, H={H}, W={W}, d_state={N})")
        offset = _c(offset)
        w_in = _need(w_in, (4 * K, K), "in_proj.weight")
        conv_w = _need(conv_w.reshape(Dn, -1), (Dn, 4), "conv1d.weight")
        conv_b = _need(conv_b, (Dn,), "conv1d.bias") if conv_b is not None else None
        w_x = _need(w_x, (1 + 2 * N, Dn), "x_proj.weight")
        w_dt = _need(w_dt.reshape(-1), (Dn,), "dt_proj.weight")
        dt_bias = _need(dt_bias, (Dn,), "dt_proj.bias") if dt_bias is not None else None
        A = _need(A, (Dn, N), "A")
        D = _need(D, (Dn,), "D") if D is not None else None
        w_out = _need(w_out, (K, Dn), "out_proj.weight")
        if altho.numel() != 1 or altho.dtype != torch.float32:
            raise RuntimeError("mamba_small_fused: altho must hold one float32 element")
        al = _c(altho.reshape(1))
        parts = L.mmu_mamba_small_parts(B, K, H, W, N, 0)
        y = torch.empty((parts, B, K, H, W), device=offset.device, dtype=torch.float32)
        p = _lib.MambaSmallParams()
        p.batch, p.height, p.width, p.taps, p.dstate, p.parts, p.extend_scope = B, H, W, K, N, parts, float(scope)
        p.offset, p.in_proj_weight, p.conv_weight, p.conv_bias = offset.data_ptr(), w_in.data_ptr(), conv_w.data_ptr(), \
            _lib.ptr(conv_b)
        p.x_proj_weight, p.dt_proj_weight, p.dt_bias, p.A, p.D = w_x.data_ptr(), w_dt.data_ptr(), _lib.ptr(dt_bias), \
            A.data_ptr(), _lib.ptr(D)
        p.out_proj_weight, p.altho, p.y = w_out.data_ptr(), al.data_ptr(), y.data_ptr()
        with torch.cuda.device(offset.device):
            _lib.check(L.mmu_mamba_small_fwd(p, _lib.stream_of(offset)))
        ctx.save_for_backward(offset, w_in, conv_w, conv_b, w_x, w_dt, dt_bias, A, D, w_out, al)
        ctx.scope, ctx.parts = float(scope), parts
        ctx.altho_shape = altho.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        offset, w_in, conv_w, conv_b, w_x, w_dt, dt_bias, A, D, w_out, al = ctx.saved_tensors
        B, C2, H, W = offset.shape
        K = C2 // 2
        Dn, N = 2 * K, A.shape[1]
        L = _lib.lib()
        if tuple(dy.shape) != (ctx.parts, B, K, H, W):
            raise RuntimeError(f"mamba_small_fused: gradient shape {tuple(dy.shape)} != {(ctx.parts, B, K, H, W)}")
        # every part's gradient is the gradient of the sum: the sampler hands back a stride-0 view of it (any other
        # producer: the parts' gradients must agree, which a plain sum of the parts guarantees)
        if ctx.parts > 1 and dy.stride(0) != 0:
            raise RuntimeError("mamba_small_fused: the partial row maps may only be consumed through their sum "
                               "(morph_sample / .sum(0)); got per-part gradients")
        g = _c(dy[0].float())
        parts = L.mmu_mamba_small_parts(B, K, H, W, N, 1)     # (the backward's own split of the states)
        doff = torch.empty_like(offset)
        nv = L.mmu_mamba_small_grad_floats(K, N)
        ws = torch.empty(L.mmu_mamba_small_bwd_workspace_floats(B, K, H, W, N, parts), device=offset.device,
                         dtype=torch.float32)
        dw = torch.empty(nv, device=offset.device, dtype=torch.float32)
        p = _lib.MambaSmallParams()
        p.batch, p.height, p.width, p.taps, p.dstate, p.parts, p.extend_scope = B, H, W, K, N, parts, ctx.scope
        p.offset, p.in_proj_weight, p.conv_weight, p.conv_bias = offset.data_ptr(), w_in.data_ptr(), conv_w.data_ptr(), \
            _lib.ptr(conv_b)
        p.x_proj_weight, p.dt_proj_weight, p.dt_bias, p.A, p.D = w_x.data_ptr(), w_dt.data_ptr(), _lib.ptr(dt_bias), \
            A.data_ptr(), _lib.ptr(D)
        p.out_proj_weight, p.altho = w_out.data_ptr(), al.data_ptr()
        p.dy, p.doffset, p.workspace, p.dweights = g.data_ptr(), doff.data_ptr(), ws.data_ptr(), dw.data_ptr()
        with torch.cuda.device(offset.device):
            _lib.check(L.mmu_mamba_small_bwd(p, _lib.stream_of(offset)))
        # the layout of dweights (include/mmunet_amd.h)
        sizes = (("w_in", 4 * K * K, (4 * K, K)), ("conv_w", Dn * 4, (Dn, 1, 4)), ("conv_b", Dn, (Dn,)),
                 ("w_x", (1 + 2 * N) * Dn, (1 + 2 * N, Dn)), ("w_dt", Dn, (Dn, 1)), ("dt_bias", Dn, (Dn,)),
                 ("A", Dn * N, (Dn, N)), ("D", Dn, (Dn,)), ("w_out", K * Dn, (K, Dn)), ("altho", 1, ctx.altho_shape))
        out, o = {}, 0
        for name, n, shape in sizes:
            out[name] = dw[o:o + n].view(shape)
            o += n
        return (doff, out["w_in"], out["conv_w"], out["conv_b"] if conv_b is not None else None, out["w_x"],
                out["w_dt"], out["dt_bias"] if dt_bias is not None else None, out["A"],
                out["D"] if D is not None else None, out["w_out"], out["altho"], None)


def mamba_rows(offset, mamba, altho, extend_scope, A=None, combine=True):
    """Row coordinates of MMConv's K taps through the fused kernels.  ``A``: ``-exp(A_log)`` when the caller has it
    (``mamba_simple.neg_exp``: one batched launch for all blocks of a model), computed here otherwise.
    ``combine=False`` returns the kernel's ``(parts, B, K, H, W)`` partial maps as they are -- ``morph_sample`` takes
    them in that form and adds them up while it reads them; ``combine=True`` adds them here (one more launch)."""
    m = mamba
    if A is None:
        A = -torch.exp(m.A_log.float())
    y = MambaSmallFusedFn.apply(offset, m.in_proj.weight, m.conv1d.weight, m.conv1d.bias, m.x_proj.weight,
                                m.dt_proj.weight, m.dt_proj.bias, A, m.D, m.out_proj.weight, altho, extend_scope)
    if not combine:
        return y
    return y[0] if y.shape[0] == 1 else y.sum(0)
