"""``conv3x3_small`` -- ``nn.Conv2d(Cin, CO, 3, padding=1)`` with CO in {1, 2, 6, 8} as direct HIP kernels.

MMConv's ``offset_conv`` (Cin -> 2K = 6 channels, src/UM_Net/MMUNet.py:46,250) runs 44 times per MM-UNet
forward.  Six output channels leave a matrix core nothing to do; MIOpen's Winograd / implicit-GEMM kernels
take 60 us forward and 140 us backward for [8, 64, 128, 128] (5.3 ms per training step in total).  Contiguous
NCHW; the input (and its gradient) float32 or bfloat16 -- bf16 activations under autocast are read as they are, the
arithmetic, the weights and the 6-channel output are float32 (more exact than autocast's bf16 convolution, and
the output feeds a GroupNorm that autocast runs in float32 anyway); anything else is the caller's ``F.conv2d``.
"""
import os

import torch

from . import _lib, deferred

SUPPORTED_CO = (1, 2, 6, 8)
WEIGHT_GRAD_NATIVE = True    # False: weight / bias gradient from ATen (MIOpen); tests cover both


ENABLED = True   # False: callers keep their nn.Conv2d call (fused_paths.plain_aten, A/B tests)


def supported(x, weight):
    return (ENABLED and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and weight.dtype == torch.float32 and x.dim() == 4
            and weight.shape[2:] == (3, 3) and weight.shape[0] in SUPPORTED_CO)


# False (MMUNET_GRAD_HANDOVER=0): no gradient hand-over slots are created -- every consumer returns its input gradient to
# autograd, which adds them (the same kernels otherwise: tools/dbg/handover_vs_autograd.py compares the two at full size)
HANDOVER = os.environ.get("MMUNET_GRAD_HANDOVER", "1") != "0"


def grad_slot():
    return GradSlot() if HANDOVER else None


def shared_grad():
    return SharedGrad() if HANDOVER else None


class GradSlot:
    """Hand-over of an input gradient between the two consumers of one tensor (MMConv: the offset convolution and
    the sampler both read the block's input).  The consumer whose backward runs FIRST (the sampler: the convolution's
    output gradient depends on the sampler's d(row)) parks its input gradient here and returns None; the convolution's
    backward adds its own contribution to it in the same kernel and returns the sum -- autograd has one gradient to
    route instead of two to add (41 activation-sized adds per training step).  ``armed`` is set by the convolution's
    forward when its backward will produce an input gradient."""
    __slots__ = ("armed", "grad", "extra", "extra_ok")

    def __init__(self):
        # extra / extra_ok: a third consumer's gradient (park_extra), accepted while the sampler's backward is still to come
        self.armed, self.grad, self.extra, self.extra_ok = False, None, None, False


class _ParkExtraFn(torch.autograd.Function):
    """Identity on a THIRD consumer's view of the tensor a GradSlot's pair reads (ResidualBlock: the block input also
    feeds the residual add): its gradient is parked in ``slot.extra`` and the sampler's backward -- the first of the pair
    to run -- adds it in its gather kernel.  Applied right before the op that consumes the view, so that autograd runs this
    node before the pair's (later-created nodes first); if the pair has already run, the gradient goes the normal way."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        ctx.like = (x.dtype, tuple(x.shape))
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        slot = ctx.slot
        if slot.extra_ok and slot.extra is None and g.is_contiguous() and (g.dtype, tuple(g.shape)) == ctx.like:
            slot.extra = g
            return None, None
        return g, None


def park_extra(x, slot):
    return _ParkExtraFn.apply(x, slot) if (slot is not None and x.requires_grad and torch.is_grad_enabled()) else x


class SharedGrad:
    """The same hand-over for ANY number of consumers of one tensor (MM_Net: the CBAM edge map feeds three RCG blocks and
    the line head; the stem features feed the max-pool and CBAM).  Every consumer whose backward produces an input
    gradient calls ``join()`` in its forward; in its backward it ``take()``s what the consumers before it left (None for
    the first), writes ``own + taken`` in one kernel and hands the result to ``give()``, which parks it and returns None
    while consumers are still to come and returns the total to the last one -- autograd routes one gradient and adds
    nothing (each add it would run is a 400 MB pass at 256 x 256 x 64 x 8)."""
    __slots__ = ("pending", "grad")

    def __init__(self):
        self.pending, self.grad = 0, None

    def join(self):
        self.pending += 1

    def take(self):
        g, self.grad = self.grad, None
        return g

    def give(self, total):
        self.pending -= 1
        if self.pending > 0:
            self.grad = total
            return None
        return total


class _JoinGradFn(torch.autograd.Function):
    """Identity that makes whatever consumes its output a member of a SharedGrad without kernel support: first in line it
    parks the gradient as it is; later it adds what was parked with an ATen add (the order autograd runs MM_Net's
    branches in makes it the first)."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        slot.join()
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        parked = ctx.slot.take()
        return ctx.slot.give(g if parked is None else parked.add_(g)), None


def shared_input(x, slot):
    """``x`` for a consumer that cannot add a parked gradient itself (see SharedGrad)."""
    return _JoinGradFn.apply(x, slot) if (slot is not None and x.requires_grad and torch.is_grad_enabled()) else x


class Conv3x3SmallFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, slot=None):
        _lib.require_gpu(x, weight)
        if not supported(x, weight) or weight.shape[1] != x.shape[1] or \
                (bias is not None and (bias.dtype != torch.float32 or bias.numel() != weight.shape[0])):
            raise RuntimeError("conv3x3_small: float32 / bfloat16 NCHW input, [CO, Cin, 3, 3] float32 weight with CO in "
                               f"{SUPPORTED_CO} and a float32 bias of CO elements required")
        x = x.contiguous()
        weight = weight.contiguous()
        bias = bias.contiguous() if bias is not None else None
        B, Cin, H, W = x.shape
        CO = weight.shape[0]
        wt = weight                                           # read in place ([CO][Cin][3][3]: weight_native)
        out = torch.empty((B, CO, H, W), device=x.device, dtype=torch.float32)
        p = _lib.Conv3x3sParams()
        p.batch, p.in_channels, p.out_channels, p.height, p.width = B, Cin, CO, H, W
        p.input, p.weight_t, p.bias, p.out = x.data_ptr(), wt.data_ptr(), _lib.ptr(bias), out.data_ptr()
        p.weight_native = 1
        p.in_dtype = _lib.dtype_code(x)
        splits = _lib.lib().mmu_conv3x3_small_fwd_splits(B, Cin, H, W)
        ws = torch.empty((splits,) + tuple(out.shape), device=x.device, dtype=torch.float32) if splits > 1 else None
        p.workspace = _lib.ptr(ws)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_conv3x3_small_fwd(p, _lib.stream_of(x)))
        ctx.save_for_backward(x, wt)
        ctx.has_bias = bias is not None
        ctx.may_defer = deferred.may_defer(weight, bias)
        ctx.wshape = tuple(weight.shape)
        ctx.slot = slot
        if isinstance(slot, SharedGrad):
            if ctx.needs_input_grad[0]:
                slot.join()
            else:
                ctx.slot = None
        elif slot is not None:
            slot.armed = bool(ctx.needs_input_grad[0])
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wt = ctx.saved_tensors
        B, Cin, H, W = x.shape
        CO = ctx.wshape[0]
        g = dout.float().contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = dw = db = None
        if need_x:
            parked = None
            if isinstance(ctx.slot, SharedGrad):
                parked = ctx.slot.take()
            elif ctx.slot is not None:
                if ctx.slot.grad is not None:
                    parked, ctx.slot.grad = ctx.slot.grad, None
                # from here on nobody collects a parked gradient any more: a producer whose backward runs AFTER this one
                # (another node order, a second autograd.grad over part of the graph) must return its gradient the normal
                # way instead of parking it for good
                ctx.slot.armed = False
            if parked is not None:
                if parked.shape != x.shape or parked.dtype != x.dtype or not parked.is_contiguous():
                    raise RuntimeError("conv3x3_small: parked input gradient does not match the input")
            dx = parked if parked is not None else torch.empty_like(x)     # (in place on the parked gradient)
            p = _lib.Conv3x3sParams()
            p.batch, p.in_channels, p.out_channels, p.height, p.width = B, Cin, CO, H, W
            p.input, p.weight_t, p.dout, p.dinput = x.data_ptr(), wt.data_ptr(), g.data_ptr(), dx.data_ptr()
            p.weight_native = 1
            p.dinput_addend = _lib.ptr(parked)
            p.in_dtype = _lib.dtype_code(x)
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().mmu_conv3x3_small_bwd(p, _lib.stream_of(x)))
            if isinstance(ctx.slot, SharedGrad):
                dx = ctx.slot.give(dx)
        if need_w or need_b:
            if WEIGHT_GRAD_NATIVE:
                dw = torch.empty(ctx.wshape, device=x.device, dtype=torch.float32)
                db = torch.empty(CO, device=x.device, dtype=torch.float32) if need_b else None
                p = _lib.Conv3x3sParams()
                p.batch, p.in_channels, p.out_channels, p.height, p.width = B, Cin, CO, H, W
                p.input, p.weight_t, p.dout = x.data_ptr(), wt.data_ptr(), g.data_ptr()
                p.dweight, p.dbias = dw.data_ptr(), _lib.ptr(db)
                p.in_dtype = _lib.dtype_code(x)
                nws = _lib.lib().mmu_conv3x3_small_wgrad_workspace_floats(B, Cin, CO, H, W)
                ws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws else None
                p.workspace = _lib.ptr(ws)
                with torch.cuda.device(x.device), deferred.guard(ctx.may_defer):
                    _lib.check(_lib.lib().mmu_conv3x3_small_bwd(p, _lib.stream_of(x)))
                deferred.keep(ws)   # (deferred.Scope: the sum over the row blocks runs later)
            else:
                # ATen / MIOpen weight gradient (kept for comparison: its implicit-GEMM kernel plus layout
                # transposes take 50-110 us where the row-walking kernel needs a fraction of that)
                w = wt
                _, dw, db = torch.ops.aten.convolution_backward(
                    g, x.float(), w, [CO] if need_b else None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                    [False, bool(need_w), bool(need_b)])
        return dx, dw, db, None


def conv3x3_small(x, weight, bias=None, slot=None):
    return Conv3x3SmallFn.apply(x, weight, bias, slot)
