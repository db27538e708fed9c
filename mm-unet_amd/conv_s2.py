"""Stride-2 dense convolutions on the bf16 matrix cores with float32 accuracy (csrc/conv_s2_mfma.hip).

``nn.Conv2d(Cin, Cout, 3 | 4, stride=2, padding=1)`` -- the opening convolution of MM-UNet's down-sampling ResidualBlocks
(src/UM_Net/MMUNet.py:439-452) and RCG's ``downsample`` (:375) -- and ``nn.ConvTranspose2d(Cin, Cout, 4, stride=2,
padding=1)`` -- RCG's ``upsample`` (:360-362): forward, input gradient (each is the other's kernel) and weight gradient.
float32 NCHW, even H and W, channel multiples as the kernels need them (``*_supported``); anything else is the caller's
module call (ATen / MIOpen).  No CPU path.
"""
import os

import torch

from . import _lib, deferred

ENABLED = os.environ.get("MMUNET_CONV_S2_MFMA", "1") != "0"   # False: the modules' own calls (A/B runs, tests)


def _run(transposed, inp, weight, bias, cin, cout, k, out_hw):
    B, _, H, W = inp.shape
    out = torch.empty((B, cout, out_hw[0], out_hw[1]), device=inp.device, dtype=inp.dtype)
    L = _lib.lib()
    ws = torch.empty(L.mmu_conv_s2_workspace_bytes(cin, cout), device=inp.device, dtype=torch.uint8)
    p = _lib.ConvS2Params()
    p.batch, p.in_channels, p.out_channels, p.in_height, p.in_width = B, cin, cout, H, W
    p.out_height, p.out_width, p.kernel = out_hw[0], out_hw[1], k
    p.input, p.weight, p.bias, p.out, p.workspace = inp.data_ptr(), weight.data_ptr(), _lib.ptr(bias), out.data_ptr(), \
        ws.data_ptr()
    p.io_dtype = _lib.dtype_code(inp)    # bfloat16 activations (autocast): the kernels' XB forms, float32 weights
    with torch.cuda.device(inp.device):
        _lib.check((L.mmu_conv_s2_transposed_mfma if transposed else L.mmu_conv_s2_mfma)(p, _lib.stream_of(inp)))
    return out


def _wgrad(high, low, k):
    """d weight [C_low][C_high][k][k] from the high-resolution tensor (the strided conv's input / the transposed conv's
    output gradient) and the low-resolution one (the strided conv's output gradient / the transposed conv's input)."""
    B, ch, H, W = high.shape
    cl, Ho, Wo = low.shape[1], low.shape[2], low.shape[3]
    dw = torch.empty((cl, ch, k, k), device=high.device, dtype=torch.float32)
    L = _lib.lib()
    with torch.cuda.device(high.device):   # the workspace size follows the CU count of the device that runs the kernel
        nws = L.mmu_conv_s2_wgrad_workspace_floats(B, ch, cl, Ho, Wo)
    ws = torch.empty(nws, device=high.device, dtype=torch.float32)
    p = _lib.ConvS2Params()
    p.batch, p.in_channels, p.out_channels, p.in_height, p.in_width = B, ch, cl, H, W
    p.out_height, p.out_width, p.kernel = Ho, Wo, k
    p.input, p.weight, p.out, p.workspace = high.data_ptr(), low.data_ptr(), dw.data_ptr(), ws.data_ptr()
    p.io_dtype = _lib.dtype_code(high)   # bfloat16: both operands exact bf16 values, one MFMA per product; dW float32
    with torch.cuda.device(high.device):
        _lib.check(L.mmu_conv_s2_wgrad_mfma(p, _lib.stream_of(high)))
    deferred.keep(ws)    # (inside a deferred.Scope the sum over the workgroups' partials runs later)
    return dw


def _bias_grad(g):
    """sum over (batch, height, width) of a float32 NCHW gradient: a streaming pass (csrc/sum_parts.hip) where the shape
    allows it, ATen's reduction otherwise."""
    B, C, H, W = g.shape
    if (H * W) % 4 != 0 or g.data_ptr() % 16 != 0 or not g.is_contiguous():
        return g.sum(dim=(0, 2, 3))
    ws = torch.empty(B * C, device=g.device, dtype=torch.float32)
    out = torch.empty(C, device=g.device, dtype=torch.float32)
    with torch.cuda.device(g.device):
        _lib.check(_lib.lib().mmu_channel_sum(g.data_ptr(), B, C, H * W, ws.data_ptr(), out.data_ptr(), _lib.stream_of(g)))
    deferred.keep(ws)    # (inside a deferred.Scope the sum over the batch runs later)
    return out


def _wgrad_ok(c_high, c_low, wo, low):
    return c_high % 64 == 0 and c_low % 64 == 0 and wo % 4 == 0 and low.data_ptr() % 16 == 0


LOWP = os.environ.get("MMUNET_CONV_S2_LOWP", "1") != "0"   # "0": under bf16 autocast the module call (MIOpen) stays


def _io_ok(x, as_bf16=False):
    """float32 activations outside autocast, or bfloat16 ones under bf16 autocast (the kernels' XB forms; maps within 32-bit
    byte offsets of the buffer addressing).  ``as_bf16``: judge a float32 tensor as its bfloat16 cast."""
    if x.dtype == torch.float32 and not as_bf16:
        return not torch.is_autocast_enabled()
    return (LOWP and (x.dtype == torch.bfloat16 or as_bf16) and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16 and 4 * 64 * 4 * x.shape[2] * x.shape[3] < 2 ** 31)


def _f32c(x, *ts):
    if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("conv_s2: float32 or bfloat16 GPU activations required")
    for t in ts:
        if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
            raise RuntimeError("conv_s2: float32 GPU weights / bias required")


def conv_supported(x, weight, as_bf16=False):
    """Conv2d(k, stride 2, padding 1) forward through the kernel: weight [Cout, Cin, k, k], k in (3, 4)."""
    return (ENABLED and x.is_cuda and x.dim() == 4 and _io_ok(x, as_bf16) and weight.dtype == torch.float32
            and weight.dim() == 4 and weight.shape[2] == weight.shape[3] and weight.shape[2] in (3, 4)
            and weight.shape[1] == x.shape[1] and weight.shape[1] % 16 == 0 and weight.shape[0] % 64 == 0
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)


def convt_supported(x, weight, as_bf16=False):
    """ConvTranspose2d(4, stride 2, padding 1) forward through the kernel: weight [Cin, Cout, 4, 4]."""
    return (ENABLED and x.is_cuda and x.dim() == 4 and _io_ok(x, as_bf16) and weight.dtype == torch.float32
            and weight.dim() == 4 and tuple(weight.shape[2:]) == (4, 4) and weight.shape[0] == x.shape[1]
            and weight.shape[0] % 16 == 0 and weight.shape[1] % 64 == 0)


class ConvS2Fn(torch.autograd.Function):
    """``F.conv2d(x, weight, bias, stride=2, padding=1)``."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _lib.require_gpu(x, weight)
        _f32c(x, weight, bias)
        if not conv_supported(x, weight) or (bias is not None and bias.numel() != weight.shape[0]):
            raise RuntimeError("conv_s2: float32 (bfloat16 under autocast) NCHW input with even H, W; [Cout, Cin, k, k] weight with k in (3, 4), "
                               "Cin % 16 == 0, Cout % 64 == 0; bias of Cout elements")
        x, weight = x.contiguous(), weight.contiguous()
        bias = bias.contiguous() if bias is not None else None
        cout, cin, k = weight.shape[0], weight.shape[1], weight.shape[2]
        ho, wo = (x.shape[2] + 2 - k) // 2 + 1, (x.shape[3] + 2 - k) // 2 + 1
        out = _run(False, x, weight, bias, cin, cout, k, (ho, wo))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.may_defer = deferred.may_defer(weight, bias)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight = ctx.saved_tensors
        cout, cin, k = weight.shape[0], weight.shape[1], weight.shape[2]
        lowp = x.dtype != torch.float32          # bf16 activations (autocast): the kernels' XB forms
        g = (dout.to(x.dtype) if lowp else dout.float()).contiguous()
        wa = lambda: weight.to(g.dtype) if lowp else weight   # noqa: E731  (the ATen routes below only)
        dx = dw = db = None
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[0]:
            if cout % 16 == 0 and cin % 64 == 0:     # the transposed kernel on the weight as it is
                dx = _run(True, g, weight, None, cout, cin, k, (x.shape[2], x.shape[3]))
            else:
                dx = torch.ops.aten.convolution_backward(g, x, wa(), None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            if _wgrad_ok(cin, cout, g.shape[3], g):
                with deferred.guard(ctx.may_defer):
                    dw = _wgrad(x, g, k)
            else:
                dw = torch.ops.aten.convolution_backward(g, x, wa(), None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1].float()
        if need_b:
            with deferred.guard(ctx.may_defer):
                db = g.sum(dim=(0, 2, 3), dtype=torch.float32) if lowp else _bias_grad(g)
        return dx, dw, db


class ConvT2Fn(torch.autograd.Function):
    """``F.conv_transpose2d(x, weight, bias, stride=2, padding=1)`` for a 4 x 4 kernel (output = 2 x input)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _lib.require_gpu(x, weight)
        _f32c(x, weight, bias)
        if not convt_supported(x, weight) or (bias is not None and bias.numel() != weight.shape[1]):
            raise RuntimeError("conv_s2 (transposed): float32 (bfloat16 under autocast) NCHW input; [Cin, Cout, 4, 4] weight with Cin % 16 == 0, "
                               "Cout % 64 == 0; bias of Cout elements")
        x, weight = x.contiguous(), weight.contiguous()
        bias = bias.contiguous() if bias is not None else None
        cin, cout = weight.shape[0], weight.shape[1]
        out = _run(True, x, weight, bias, cin, cout, 4, (2 * x.shape[2], 2 * x.shape[3]))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.may_defer = deferred.may_defer(weight, bias)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight = ctx.saved_tensors
        cin, cout = weight.shape[0], weight.shape[1]
        lowp = x.dtype != torch.float32
        g = (dout.to(x.dtype) if lowp else dout.float()).contiguous()
        wa = lambda: weight.to(g.dtype) if lowp else weight   # noqa: E731  (the ATen routes below only)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if cout % 16 == 0 and cin % 64 == 0:     # the strided kernel on the weight as it is: [Cin_T][Cout_T] = [out][in]
                dx = _run(False, g, weight, None, cout, cin, 4, (x.shape[2], x.shape[3]))
            else:
                dx = torch.nn.functional.conv2d(g, wa(), None, stride=2, padding=1)
        if ctx.needs_input_grad[1]:
            if _wgrad_ok(cout, cin, x.shape[3], x):
                with deferred.guard(ctx.may_defer):
                    dw = _wgrad(g, x, 4)            # [C_low = Cin_T][C_high = Cout_T][4][4]
            else:
                dw = torch.ops.aten.convolution_backward(g, x, wa(), None, [2, 2], [1, 1], [1, 1], True, [0, 0], 1,
                                                         [False, True, False])[1].float()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            with deferred.guard(ctx.may_defer):
                db = g.sum(dim=(0, 2, 3), dtype=torch.float32) if lowp else _bias_grad(g)
        return dx, dw, db


def conv_s2(x, weight, bias=None):
    return ConvS2Fn.apply(x, weight, bias)


def conv_transpose_s2(x, weight, bias=None):
    return ConvT2Fn.apply(x, weight, bias)


def _autocast_view(x):
    """What the layer would see under bf16 autocast: float32 activations are cast to bfloat16 (as F.conv2d's autocast rule
    does) so that the kernels' bf16 forms take them; anything else is returned as it is."""
    if (LOWP and x.dtype == torch.float32 and x.is_cuda and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return x.to(torch.bfloat16)
    return x


def module_supported(m, x):
    """True when ``m`` is a stride-2 ``nn.Conv2d`` / ``nn.ConvTranspose2d`` these kernels cover for the input ``x``."""
    # (float32 activations under bf16 autocast are judged as the bfloat16 tensor module_call will make of them)
    cast = (LOWP and x.dtype == torch.float32 and x.is_cuda and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16)
    if isinstance(m, torch.nn.ConvTranspose2d):
        return (m.kernel_size == (4, 4) and m.stride == (2, 2) and m.padding == (1, 1) and m.output_padding == (0, 0)
                and m.dilation == (1, 1) and m.groups == 1 and convt_supported(x, m.weight, cast)
                and (m.bias is None or m.bias.dtype == torch.float32))
    if isinstance(m, torch.nn.Conv2d):
        return (m.kernel_size in ((3, 3), (4, 4)) and m.stride == (2, 2) and m.padding == (1, 1) and m.dilation == (1, 1)
                and m.groups == 1 and m.padding_mode == "zeros" and conv_supported(x, m.weight, cast)
                and (m.bias is None or m.bias.dtype == torch.float32))
    return False


def module_call(m, x):
    """``m(x)`` through the matrix-core kernels when :func:`module_supported`, the module itself otherwise."""
    if not module_supported(m, x):
        return m(x)
    x = _autocast_view(x)
    if isinstance(m, torch.nn.ConvTranspose2d):
        return ConvT2Fn.apply(x, m.weight, m.bias)
    return ConvS2Fn.apply(x, m.weight, m.bias)
