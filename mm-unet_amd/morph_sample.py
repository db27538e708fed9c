"""``morph_sample`` -- MMConv's deformable sampling as one HIP op (fwd + bwd).

Stands where MMConv.get_interpolated_feature stands in the reference (src/UM_Net/MMUNet.py:196-242):
coordinate clamp + scaling to [-1, 1] + grid construction + ``F.grid_sample(bilinear, zeros,
align_corners=True)``.  In MMConv the sampling column of tap k at pixel (h, w) is the integer
``w + k - K//2`` (MMUNet.py:141-151) and only the row coordinate is learned, so the op takes the row
coordinate map alone:

    out[b, c, h*K + k, w] = lerp(input[b, c, floor(yc), col], input[b, c, floor(yc)+1, col], frac(yc))
    yc = clamp(y[b, k, h, w], 0, H-1),  col = clamp(w + k - K//2, 0, W-1)

Gradients: d input (a gather over the reachable sources; float atomics only for far outliers) and d y
(zero where y was clamped, like torch.clamp).  The input (and its gradient) may be bfloat16 -- activations under
autocast are read as they are; coordinates, interpolation and the samples are float32 (``F.grid_sample`` is on
autocast's fp32 list, so the reference computes this step in fp32 under autocast as well).

``tokens_last=True`` returns the samples as the ``(C*K, B*H*W)`` matrix ``[c][k][b][h][w]``: the K x 1 /
stride K x 1 ``dsc_conv_x`` that consumes them (MMUNet.py:262) is then ``weight.view(Cout, Cin*K) @ samples``
-- one GEMM instead of an implicit-GEMM convolution with layout transposes around it.
"""
import torch

from . import _lib

ENABLED = True   # False: MMConv samples with F.grid_sample (its reference-shaped method; fused_paths.plain_aten)


class MorphSampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, y, tokens_last=False, slot=None):
        _lib.require_gpu(input, y)
        # y may come as (parts, B, K, H, W): partial row maps whose sum is the row map (mamba_small_fused); the kernel
        # adds them up while it reads them and keeps the total for the backward
        parts = y.shape[0] if y.dim() == 5 else 0
        ys = y.shape[1:] if parts else y.shape
        if input.dim() != 4 or len(ys) != 4 or ys[0] != input.shape[0] or ys[2:] != input.shape[2:]:
            raise RuntimeError("morph_sample: input must be (B, C, H, W) and y (B, K, H, W) [or (parts, B, K, H, W)]")
        if ys[1] % 2 != 1:
            raise RuntimeError("morph_sample: the number of taps K must be odd")
        x = (input if input.dtype in (torch.float32, torch.bfloat16) else input.float()).contiguous()
        yin = y.float().contiguous()
        yy = torch.empty(tuple(ys), device=yin.device, dtype=torch.float32) if parts > 1 else (yin[0] if parts else yin)
        B, C, H, W = x.shape
        K = yy.shape[1]
        shape = (C * K, B * H * W) if tokens_last else (B, C, H * K, W)
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
        p = _lib.MorphParams()
        p.batch, p.channels, p.height, p.width, p.taps = B, C, H, W, K
        p.out_layout = int(tokens_last)
        p.input, p.y, p.out = x.data_ptr(), yin.data_ptr(), out.data_ptr()
        if parts > 1:
            p.y_parts, p.y_sum = parts, yy.data_ptr()
        p.in_dtype = _lib.dtype_code(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_morph_sample_fwd(p, _lib.stream_of(x)))
        ctx.save_for_backward(x, yy)
        ctx.parts = parts
        ctx.in_dtype, ctx.y_dtype, ctx.tokens_last = input.dtype, y.dtype, bool(tokens_last)
        ctx.slot = slot
        if slot is not None:   # this call's backward can add a third consumer's gradient of the input (park_extra)
            slot.extra_ok = bool(ctx.needs_input_grad[0]) and x.dtype == input.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        x, yy = ctx.saved_tensors
        B, C, H, W = x.shape
        K = yy.shape[1]
        g = dout.float().contiguous()
        dinput = torch.empty_like(x)
        dy = torch.empty_like(yy)
        p = _lib.MorphParams()
        p.batch, p.channels, p.height, p.width, p.taps = B, C, H, W, K
        p.out_layout = int(ctx.tokens_last)
        p.input, p.y, p.dout = x.data_ptr(), yy.data_ptr(), g.data_ptr()
        p.dinput, p.dy = dinput.data_ptr(), dy.data_ptr()
        p.in_dtype = _lib.dtype_code(x)
        slot = ctx.slot
        extra = None
        if slot is not None:
            slot.extra_ok = False      # (from here on a late park_extra gradient goes the normal way)
        if slot is not None and slot.extra is not None:    # a residual connection's gradient of x (conv3x3_small.park_extra)
            extra, slot.extra = slot.extra, None
            if extra.shape != x.shape or extra.dtype != x.dtype or not extra.is_contiguous():
                raise RuntimeError("morph_sample: parked residual gradient does not match the input")
            p.dinput_addend = extra.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_morph_sample_bwd(p, _lib.stream_of(x)))
        dinput = dinput.to(ctx.in_dtype)
        # conv3x3_small.GradSlot: the offset convolution's backward (which needs this call's d(row) and therefore
        # runs after it) adds its own input gradient to this one and returns the sum
        if slot is not None and slot.armed and ctx.needs_input_grad[1] and slot.grad is None and x.dtype == ctx.in_dtype:
            slot.grad = dinput
            dinput = None
        dy = dy.to(ctx.y_dtype)
        if ctx.parts:   # every partial map receives the gradient of their sum: a stride-0 view, nothing is copied
            dy = dy.unsqueeze(0).expand(ctx.parts, *dy.shape)
        return dinput, dy, None, None


def morph_sample(input, y, tokens_last=False, slot=None):
    """input (B, C, H, W), y (B, K, H, W) row coordinates in pixels -> (B, C, H*K, W), or the
    (C*K, B*H*W) matrix ``[c][k][b][h][w]`` with ``tokens_last=True``.  ``slot``: a conv3x3_small.GradSlot shared
    with the offset convolution that reads the same input (see there)."""
    return MorphSampleFn.apply(input, y, tokens_last, slot)
