"""Fused glue of MMConv around its K-channel Mamba (HIP kernels in csrc/morph_coords.hip).

``zigzag_inproj(offset, in_proj_weight)``  -> xz as a (B, 4K, L) view of a tokens-last [4K][B][L] buffer:
    the two-row zig-zag flatten of ``offset[:, :K]`` (MMUNet.py:68-93,178-180) followed by ``Mamba.in_proj``
    (mamba_simple.py:201-205).
``coords_outproj(offset, out_z, out_proj_weight, altho, extend_scope)`` -> y (B, K, H, W):
    ``Mamba.out_proj`` (mamba_simple.py:365), the inverse zig-zag (MMUNet.py:95-121,182-183) and
    ``y = clamp(softplus(altho), min=.01) * y_keep + row + extend_scope * cumsum-from-centre(offset)``
    (MMUNet.py:138-188).

float32, K in {1, 3}; anything else takes MMConv's un-fused path.
"""
import torch

from . import _lib, deferred


ENABLED = True   # False: MMConv builds its coordinate map with tensor ops (fused_paths.plain_aten)


def supported(offset, K):
    return ENABLED and offset.is_cuda and offset.dtype == torch.float32 and K in (1, 3)


def _params(offset, K, scope=1.0):
    B, _, H, W = offset.shape
    p = _lib.CoordsParams()
    p.batch, p.height, p.width, p.taps = B, H, W, K
    p.extend_scope = float(scope)
    p.offset = offset.data_ptr()
    return p


def _dbl(t, rows, B, L):
    """(B, rows, L) tensor -> its [rows][B][L] tokens-last buffer (copy only if it is not laid out so)."""
    v = t.permute(1, 0, 2)
    return v if v.is_contiguous() else v.contiguous()


def _check_offset(offset, rows, w, what):
    """The kernels read raw float32 pointers: everything is brought to contiguous float32 here and shapes are
    checked before a pointer is taken (a bf16 buffer read as float32 is an out-of-bounds read: DESIGN.md 7.1)."""
    _lib.require_gpu(offset, w)
    if offset.dim() != 4 or offset.shape[1] % 2 != 0 or offset.shape[1] // 2 not in (1, 3):
        raise RuntimeError(f"{what}: offset must be (B, 2K, H, W) with K in (1, 3)")
    K = offset.shape[1] // 2
    if tuple(w.shape) != rows(K):
        raise RuntimeError(f"{what}: projection weight shape {tuple(w.shape)} != {rows(K)}")
    return offset.float().contiguous(), w.float().contiguous()


class OffsetGradSlot:
    """Hand-over of d(offset) between the two consumers of MMConv's offsets: ``coords_outproj`` (whose backward runs
    first -- it depends on nothing of the Mamba chain) parks its share here and returns None, ``zigzag_inproj``'s backward
    adds its own share in the same kernel and returns the sum: one gradient for autograd to route instead of two to add
    (an elementwise launch per MMConv backward).  Armed by ``zigzag_inproj``'s forward when its backward will run."""
    __slots__ = ("armed", "grad")

    def __init__(self):
        self.armed, self.grad = False, None


def _bwd_workspace(offset, K):
    B, _, H, W = offset.shape
    return torch.empty(_lib.lib().mmu_coords_bwd_workspace_floats(B, H, W, K), device=offset.device, dtype=torch.float32)


class ZigzagInProjFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, offset, w_in, slot=None):
        ctx.off_dtype = offset.dtype
        offset, w = _check_offset(offset, lambda K: (4 * K, K), w_in, "zigzag_inproj")
        B, C2, H, W = offset.shape
        K = C2 // 2
        buf = torch.empty((4 * K, B, H * W), device=offset.device, dtype=torch.float32)
        p = _params(offset, K)
        p.in_proj_weight, p.xz = w.data_ptr(), buf.data_ptr()
        with torch.cuda.device(offset.device):
            _lib.check(_lib.lib().mmu_zigzag_inproj_fwd(p, _lib.stream_of(offset)))
        ctx.save_for_backward(offset, w)
        ctx.w_dtype = w_in.dtype
        ctx.may_defer = deferred.may_defer(w)
        ctx.slot = slot
        if slot is not None:
            slot.armed = bool(ctx.needs_input_grad[0]) and ctx.off_dtype == torch.float32
        return buf.permute(1, 0, 2)

    @staticmethod
    def backward(ctx, dxz):
        offset, w = ctx.saved_tensors
        B, C2, H, W = offset.shape
        K = C2 // 2
        g = _dbl(dxz.float(), 4 * K, B, H * W)
        slot = ctx.slot
        parked = None
        if slot is not None:
            if slot.grad is not None:
                parked, slot.grad = slot.grad, None
            slot.armed = False   # a producer that runs after this node hands its gradient to autograd (see conv3x3_small)
        doff = parked if parked is not None else torch.empty_like(offset)
        dw = torch.empty_like(w)
        p = _params(offset, K)
        p.in_proj_weight, p.dxz, p.doffset, p.din_proj_weight = w.data_ptr(), g.data_ptr(), doff.data_ptr(), \
            dw.data_ptr()
        p.accumulate_doffset = int(parked is not None)
        ws = _bwd_workspace(offset, K)     # per-block partials: ordered sums (no atomics, no zero fill), deferrable
        p.workspace = ws.data_ptr()
        with torch.cuda.device(offset.device), deferred.guard(ctx.may_defer):
            _lib.check(_lib.lib().mmu_zigzag_inproj_bwd(p, _lib.stream_of(offset)))
        deferred.keep(ws)
        return doff.to(ctx.off_dtype), dw.to(ctx.w_dtype), None


class CoordsOutProjFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, offset, out_z, w_out, altho, scope, slot=None):
        ctx.off_dtype, ctx.oz_dtype = offset.dtype, out_z.dtype
        offset, w = _check_offset(offset, lambda K: (K, 2 * K), w_out, "coords_outproj")
        if altho.numel() != 1:
            raise RuntimeError("coords_outproj: altho must hold one element")
        al = altho.float().reshape(1).contiguous()
        B, C2, H, W = offset.shape
        K = C2 // 2
        if tuple(out_z.shape) != (B, 2 * K, H * W):
            raise RuntimeError(f"coords_outproj: out_z shape {tuple(out_z.shape)} != {(B, 2 * K, H * W)}")
        oz = _dbl(out_z.float(), 2 * K, B, H * W)
        y = torch.empty((B, K, H, W), device=offset.device, dtype=torch.float32)
        p = _params(offset, K, scope)
        p.out_proj_weight, p.altho, p.out_z, p.y = w.data_ptr(), al.data_ptr(), oz.data_ptr(), y.data_ptr()
        with torch.cuda.device(offset.device):
            _lib.check(_lib.lib().mmu_coords_outproj_fwd(p, _lib.stream_of(offset)))
        ctx.save_for_backward(offset, oz, w, al)
        ctx.scope, ctx.w_dtype, ctx.a_shape, ctx.a_dtype = scope, w_out.dtype, altho.shape, altho.dtype
        ctx.may_defer = deferred.may_defer(w, al)
        ctx.slot = slot
        return y

    @staticmethod
    def backward(ctx, dy):
        offset, oz, w, al = ctx.saved_tensors
        B, C2, H, W = offset.shape
        K = C2 // 2
        g = dy.float().contiguous()
        doff = torch.empty_like(offset)
        doz = torch.empty_like(oz)
        # d(out_proj weight) and d(altho) in one allocation: the launcher clears both accumulation targets in one pass
        wa = torch.empty(w.numel() + 1, device=w.device, dtype=torch.float32)
        dw, da = wa[:w.numel()].view(w.shape), wa[w.numel():].view(al.shape)
        p = _params(offset, K, ctx.scope)
        p.out_proj_weight, p.altho, p.out_z, p.dy = w.data_ptr(), al.data_ptr(), oz.data_ptr(), g.data_ptr()
        p.doffset, p.dout_z, p.dout_proj_weight, p.daltho = doff.data_ptr(), doz.data_ptr(), dw.data_ptr(), \
            da.data_ptr()
        ws = _bwd_workspace(offset, K)
        p.workspace = ws.data_ptr()
        with torch.cuda.device(offset.device), deferred.guard(ctx.may_defer):
            _lib.check(_lib.lib().mmu_coords_outproj_bwd(p, _lib.stream_of(offset)))
        deferred.keep(ws)
        doff = doff.to(ctx.off_dtype)
        slot = ctx.slot
        if slot is not None and slot.armed and slot.grad is None and doff.dtype == torch.float32:
            slot.grad, doff = doff, None       # zigzag_inproj's backward adds its share to it and returns the sum
        return (doff, doz.permute(1, 0, 2).to(ctx.oz_dtype), dw.to(ctx.w_dtype),
                da.reshape(ctx.a_shape).to(ctx.a_dtype), None, None)


def zigzag_inproj(offset, in_proj_weight, slot=None):
    return ZigzagInProjFn.apply(offset, in_proj_weight, slot)


def coords_outproj(offset, out_z, out_proj_weight, altho, extend_scope=1.0, slot=None):
    return CoordsOutProjFn.apply(offset, out_z, out_proj_weight, altho, extend_scope, slot)
