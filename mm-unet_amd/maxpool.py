"""``max_pool3s2`` -- ``nn.MaxPool2d(kernel_size=3, stride=2, padding=1)`` (src/UM_Net/MMUNet.py:493,537): forward with
one-byte arg-max codes (ATen's choice among equal values) and a gather backward from them (csrc/maxpool.hip): no
atomics, no zero fill, bit-reproducible; ATen: 97 us forward (int64 indices), 18 + 221 us scatter backward at
[8, 64, 256, 256].
float32 contiguous NCHW on the GPU -- or, under bf16 autocast, bfloat16 maps with W % 8 == 0 and even H (read and written
natively, comparisons / gradient sums in float32); anything else is the module's own path."""
import os

import torch
import torch.nn.functional as F

from . import _lib

ENABLED = True   # False: callers use the nn.MaxPool2d module (tests compare the two)
LOWP = os.environ.get("MMUNET_MAXPOOL_LOWP", "1") != "0"   # "0": bfloat16 maps stay on the module (A/B)


def _lowp(x):
    return (LOWP and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[3] % 8 == 0 and x.shape[2] % 2 == 0
            and x.data_ptr() % 16 == 0)


def module_supported(m, x):
    return (ENABLED and isinstance(m, torch.nn.MaxPool2d) and m.kernel_size in (3, (3, 3)) and m.stride in (2, (2, 2))
            and m.padding in (1, (1, 1)) and m.dilation in (1, (1, 1)) and not m.ceil_mode and not m.return_indices
            and x.is_cuda and x.dim() == 4
            and ((x.dtype == torch.float32 and not torch.is_autocast_enabled()) or _lowp(x)))


class MaxPool3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot=None):
        _lib.require_gpu(x)
        if x.dim() != 4 or not (x.dtype == torch.float32 or _lowp(x)):
            raise RuntimeError("max_pool3s2: float32 (B, C, H, W) tensor (or bfloat16 with W % 8 == 0, even H) required")
        x = x.contiguous()
        if x.dtype != torch.float32:
            slot = None      # (the hand-over of input gradients parks float32 tensors only: autograd adds the bf16 ones)
        B, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty((B, C, OH, OW), device=x.device, dtype=x.dtype)
        codes = torch.empty((B, C, OH, OW), device=x.device, dtype=torch.uint8)   # arg-max position inside the window
        p = _lib.MaxPoolParams()
        p.planes, p.height, p.width, p.out_height, p.out_width = B * C, H, W, OH, OW
        p.input, p.out, p.codes = x.data_ptr(), out.data_ptr(), codes.data_ptr()
        p.io_dtype = _lib.dtype_code(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_maxpool3s2_fwd(p, _lib.stream_of(x)))
        ctx.save_for_backward(codes)
        ctx.in_shape, ctx.io_dtype = x.shape, x.dtype
        # slot: a conv3x3_small.SharedGrad of all consumers of x -- their input gradients leave as one
        ctx.slot = slot if (slot is not None and ctx.needs_input_grad[0]) else None
        if ctx.slot is not None:
            ctx.slot.join()
        return out

    @staticmethod
    def backward(ctx, g):
        codes, = ctx.saved_tensors
        B, C, H, W = ctx.in_shape
        g = g.to(ctx.io_dtype).contiguous()
        parked = ctx.slot.take() if ctx.slot is not None else None
        if parked is not None and (parked.shape != ctx.in_shape or parked.dtype != torch.float32 or not parked.is_contiguous()):
            raise RuntimeError("max_pool3s2: parked input gradient does not match the input")
        dx = parked if parked is not None else torch.empty(ctx.in_shape, device=g.device, dtype=ctx.io_dtype)   # (in place)
        p = _lib.MaxPoolParams()
        p.planes, p.height, p.width, p.out_height, p.out_width = B * C, H, W, g.shape[2], g.shape[3]
        p.dout, p.codes, p.dinput, p.dinput_addend = g.data_ptr(), codes.data_ptr(), dx.data_ptr(), _lib.ptr(parked)
        p.io_dtype = _lib.dtype_code(g)
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().mmu_maxpool3s2_bwd_codes(p, _lib.stream_of(g)))
        return (ctx.slot.give(dx) if ctx.slot is not None else dx), None


def max_pool3s2(x, slot=None):
    return MaxPool3s2Fn.apply(x, slot)


def pool_module(m, x, slot=None):
    """``m(x)`` for an ``nn.MaxPool2d``: the gather-backward form when :func:`module_supported`, the module otherwise.
    ``slot``: a conv3x3_small.SharedGrad shared by all consumers of ``x``."""
    return max_pool3s2(x, slot) if module_supported(m, x) else m(x)
