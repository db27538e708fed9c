"""MM_Net's stem convolution ``nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)`` (MMUNet.py:492) on the
matrix cores with float32-grade products (csrc/stem7_mfma.hip): forward and weight gradient; the image is the network's
input, so there is no input gradient (an input that requires one keeps the module)."""
import torch

from . import _lib

ENABLED = True   # False: the stem stays an nn.Conv2d call (fused_paths.plain_aten; tests)


def supported(m, x):
    return (ENABLED and isinstance(m, torch.nn.Conv2d) and m.in_channels == 3 and m.out_channels == 64
            and m.kernel_size == (7, 7) and m.stride == (2, 2) and m.padding == (3, 3) and m.dilation == (1, 1)
            and m.groups == 1 and m.bias is None and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32
            and m.weight.dtype == torch.float32 and not torch.is_autocast_enabled() and not x.requires_grad
            and x.shape[2] % 2 == 0 and x.shape[3] % 16 == 0)


def _params(x):
    p = _lib.Stem7Params()
    p.batch, p.height, p.width = x.shape[0], x.shape[2], x.shape[3]
    p.input = x.data_ptr()
    return p


class Stem7Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        _lib.require_gpu(x, weight)
        x = x.contiguous()
        w = weight.contiguous()
        B, _, H, W = x.shape
        L = _lib.lib()
        out = torch.empty((B, 64, H // 2, W // 2), device=x.device, dtype=torch.float32)
        ws = torch.empty(L.mmu_stem7_workspace_bytes(B, H, W, 0), device=x.device, dtype=torch.uint8)
        p = _params(x)
        p.weight, p.out, p.workspace = w.data_ptr(), out.data_ptr(), ws.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(L.mmu_stem7_fwd(p, _lib.stream_of(x)))
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        B, _, H, W = x.shape
        L = _lib.lib()
        dout = dout.float().contiguous()
        dw = torch.empty((64, 3, 7, 7), device=x.device, dtype=torch.float32)
        ws = torch.empty(L.mmu_stem7_workspace_bytes(B, H, W, 1), device=x.device, dtype=torch.uint8)
        p = _params(x)
        p.dout, p.dweight, p.workspace = dout.data_ptr(), dw.data_ptr(), ws.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(L.mmu_stem7_wgrad(p, _lib.stream_of(x)))
        return None, dw


def stem_conv(m, x):
    return Stem7Fn.apply(x, m.weight)
