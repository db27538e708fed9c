"""``conv3x3_mfma`` -- dense ``nn.Conv2d(Cin, Cout, 3, padding=1)`` on the bf16 matrix cores with float32 accuracy
(csrc/conv3x3_mfma.hip: implicit GEMM, LDS-staged input patch, hi/lo bf16 split, three MFMAs per product).

Used for CBAM's 3x3 64->64 convolutions (src/UM_Net/MMUNet.py:313-338) and the plain Unet's conv stack
(model.py:5-20).  Forward and input gradient run the HIP kernel (the gradient on transposed / flipped weights);
the weight gradient has its own matrix-core kernel (csrc/conv3x3_wgrad_mfma.hip, Cin % 32 == 0) and falls back to
ATen's (MIOpen) otherwise.  float32 NCHW, Cin % 16 == 0, Cout % 64 == 0, W % 4 == 0; anything else
is the caller's ``F.conv2d``.
"""
import os

import torch

from . import _lib, conv_s2, deferred

# The hi/lo bf16 split drops the lo*lo term: ~2^-16 relative per product against 2^-24 on the library's float32 path
# (tests pin 5e-5 / 1e-4 on randn data).  ENABLED = False (or MMUNET_CONV3X3_MFMA=0 in the environment) routes every
# dense 3x3 convolution -- forward, input and weight gradient -- back to ATen / MIOpen for float32-exact runs.
ENABLED = os.environ.get("MMUNET_CONV3X3_MFMA", "1") != "0"


LOWP = os.environ.get("MMUNET_CONV3X3_MFMA_LOWP", "1") != "0"   # "0": under bf16 autocast the module call (MIOpen) stays


def _lowp(x):
    """bf16 activations under bf16 autocast: the kernel's XB form (bf16 in / out, float32 weights, two MFMAs per product)."""
    return (LOWP and x.dtype == torch.bfloat16 and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16)


def supported(x, weight):
    return (ENABLED and x.is_cuda and weight.dtype == torch.float32 and x.dim() == 4
            and tuple(weight.shape[2:]) == (3, 3) and weight.shape[1] == x.shape[1] and weight.shape[1] % 16 == 0
            and weight.shape[0] % 64 == 0 and x.shape[3] % 4 == 0
            and ((x.dtype == torch.float32 and not torch.is_autocast_enabled()) or _lowp(x)))


def _run(inp, weight, bias, cin, cout, transposed):
    B, _, H, W = inp.shape
    out = torch.empty((B, cout, H, W), device=inp.device, dtype=inp.dtype)
    ws = torch.empty(_lib.lib().mmu_conv3x3_mfma_workspace_bytes(cin, cout), device=inp.device, dtype=torch.uint8)
    p = _lib.Conv3x3MfmaParams()
    p.batch, p.in_channels, p.out_channels, p.height, p.width, p.transposed = B, cin, cout, H, W, int(transposed)
    p.input, p.weight, p.bias, p.out, p.workspace = inp.data_ptr(), weight.data_ptr(), _lib.ptr(bias), out.data_ptr(), \
        ws.data_ptr()
    p.io_dtype = _lib.dtype_code(inp)
    with torch.cuda.device(inp.device):
        _lib.check(_lib.lib().mmu_conv3x3_mfma(p, _lib.stream_of(inp)))
    return out


LOWP_WGRAD = os.environ.get("MMUNET_CONV3X3_WGRAD_LOWP", "1") != "0"   # "0": bf16 weight gradients from the library
WGRAD_MFMA = True   # False: the weight gradient comes from ATen / MIOpen (tests compare the two)


def wgrad_supported(x, cout):
    return WGRAD_MFMA and x.shape[1] % 32 == 0 and cout % 64 == 0 and x.shape[3] % 4 == 0


def _wgrad(x, g, cout):
    """dW of conv2d(x, W, padding=1) from x [B, Cin, H, W] and dout g [B, Cout, H, W] (csrc/conv3x3_wgrad_mfma.hip)."""
    B, cin, H, W = x.shape
    dw = torch.empty((cout, cin, 3, 3), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):   # the workspace size follows the CU count of the device that runs the kernel
        nws = _lib.lib().mmu_conv3x3_wgrad_mfma_workspace_floats(B, cin, cout, H, W)
    ws = torch.empty(nws, device=x.device, dtype=torch.float32)
    p = _lib.Conv3x3MfmaParams()
    p.batch, p.in_channels, p.out_channels, p.height, p.width, p.transposed = B, cin, cout, H, W, 0
    p.input, p.weight, p.out, p.workspace = x.data_ptr(), g.data_ptr(), dw.data_ptr(), ws.data_ptr()
    p.io_dtype = _lib.dtype_code(x)      # bf16: x and g are exact bf16 values, one MFMA per product; dW stays float32
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mmu_conv3x3_wgrad_mfma(p, _lib.stream_of(x)))
    deferred.keep(ws)    # (inside a deferred.Scope the sum over the workgroups' partials runs later)
    return dw


class Conv3x3MfmaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _lib.require_gpu(x, weight)
        if not supported(x, weight) or (bias is not None and (bias.dtype != torch.float32
                                                             or bias.numel() != weight.shape[0] or not bias.is_cuda)):
            raise RuntimeError("conv3x3_mfma: float32 NCHW input with W % 4 == 0, [Cout, Cin, 3, 3] float32 weight "
                               "with Cin % 16 == 0 and Cout % 64 == 0, float32 bias of Cout elements required")
        x = x.contiguous()
        weight = weight.contiguous()
        bias = bias.contiguous() if bias is not None else None
        out = _run(x, weight, bias, weight.shape[1], weight.shape[0], False)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.may_defer = deferred.may_defer(weight, bias)
        ctx.lowp = x.dtype != torch.float32
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        if ctx.lowp:   # bf16 activations: input and weight gradient on the kernels' bf16 forms
            g = dout.to(torch.bfloat16).contiguous()
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                if cin % 64 == 0 and cout % 16 == 0:
                    dx = _run(g, weight, None, cout, cin, True)
                else:
                    dx = torch.ops.aten.convolution_backward(g, x, weight.to(torch.bfloat16), None, [1, 1], [1, 1], [1, 1],
                                                             False, [0, 0], 1, [True, False, False])[0]
            need_b = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1] and LOWP_WGRAD and wgrad_supported(x, cout) and g.data_ptr() % 8 == 0:
                with deferred.guard(ctx.may_defer):
                    dw = _wgrad(x, g, cout)
                db = g.sum(dim=(0, 2, 3), dtype=torch.float32) if need_b else None
            elif ctx.needs_input_grad[1] or need_b:
                _, dw, db = torch.ops.aten.convolution_backward(
                    g, x, weight.to(torch.bfloat16), [cout] if need_b else None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                    [False, bool(ctx.needs_input_grad[1]), bool(need_b)])
                dw = dw.float() if dw is not None else None
                db = db.float() if db is not None else None
            return dx, dw, db
        g = dout.float().contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if cin % 64 == 0 and cout % 16 == 0:
                dx = _run(g, weight, None, cout, cin, True)
            else:  # the transposed problem does not fit the kernel's tiling (e.g. Cin = 16): ATen
                dx = torch.ops.aten.convolution_backward(g, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and wgrad_supported(x, cout) and g.data_ptr() % 16 == 0:
            with deferred.guard(ctx.may_defer):
                dw = _wgrad(x, g, cout)
                db = conv_s2._bias_grad(g) if need_b else None   # (streaming channel sums, csrc/sum_parts.hip)
        elif ctx.needs_input_grad[1] or need_b:
            _, dw, db = torch.ops.aten.convolution_backward(
                g, x, weight, [cout] if need_b else None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                [False, bool(ctx.needs_input_grad[1]), bool(need_b)])
        return dx, dw, db


def conv3x3_mfma(x, weight, bias=None):
    return Conv3x3MfmaFn.apply(x, weight, bias)


def module_supported(m, x):
    """True when ``m`` is an ``nn.Conv2d`` that :func:`conv3x3_mfma` covers for the input ``x``."""
    return (isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)
            and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros" and supported(x, m.weight))
