"""``max_pool3s2`` -- ``nn.MaxPool2d(kernel_size=3, stride=2, padding=1)`` (src/UM_Net/MMUNet.py:493,537) with ATen's
forward (it returns the arg-max indices) and a gather backward (csrc/maxpool.hip): no atomics, no zero fill,
bit-reproducible; ATen's scatter backward takes 18 + 221 us at [8, 64, 256, 256], this one a quarter of that.
float32 contiguous NCHW on the GPU; anything else is the module's own path."""
import torch
import torch.nn.functional as F

from . import _lib

ENABLED = True   # False: callers use the nn.MaxPool2d module (tests compare the two)


def module_supported(m, x):
    return (ENABLED and isinstance(m, torch.nn.MaxPool2d) and m.kernel_size in (3, (3, 3)) and m.stride in (2, (2, 2))
            and m.padding in (1, (1, 1)) and m.dilation in (1, (1, 1)) and not m.ceil_mode and not m.return_indices
            and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and not torch.is_autocast_enabled())


class MaxPool3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.require_gpu(x)
        if x.dim() != 4 or x.dtype != torch.float32:
            raise RuntimeError("max_pool3s2: float32 (B, C, H, W) tensor required")
        x = x.contiguous()
        out, idx = F.max_pool2d(x, 3, 2, 1, return_indices=True)
        ctx.save_for_backward(idx)
        ctx.in_shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        B, C, H, W = ctx.in_shape
        g = g.float().contiguous()
        dx = torch.empty(ctx.in_shape, device=g.device, dtype=torch.float32)
        p = _lib.MaxPoolParams()
        p.planes, p.height, p.width, p.out_height, p.out_width = B * C, H, W, g.shape[2], g.shape[3]
        p.dout, p.indices, p.dinput = g.data_ptr(), idx.data_ptr(), dx.data_ptr()
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().mmu_maxpool3s2_bwd(p, _lib.stream_of(g)))
        return dx


def max_pool3s2(x):
    return MaxPool3s2Fn.apply(x)


def pool_module(m, x):
    """``m(x)`` for an ``nn.MaxPool2d``: the gather-backward form when :func:`module_supported`, the module otherwise."""
    return max_pool3s2(x) if module_supported(m, x) else m(x)
