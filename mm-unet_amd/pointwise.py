"""``conv1x1_one`` -- ``nn.Conv2d(C, 1, kernel_size=1)`` (one output channel) as streaming HIP kernels
(csrc/pointwise_one.hip): RCG's gate ``mlp`` (src/UM_Net/MMUNet.py:386-387,414) and the side outputs' ``conv2``
(MMUNet.py:346,350).  float32 contiguous NCHW, C in {16, 64}, H*W % 4 == 0; anything else is the caller's ``F.conv2d``.
"""
import os

import torch

from . import _lib, deferred

ENABLED = True   # False: callers use F.conv2d (tests compare the two)


def supported(x, weight):
    return (ENABLED and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and tuple(weight.shape[2:]) == (1, 1) and weight.shape[0] == 1 and weight.shape[1] == x.shape[1]
            and x.shape[1] in (16, 64) and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[0] < 65536
            and not torch.is_autocast_enabled())


def module_supported(m, x):
    return (isinstance(m, torch.nn.Conv2d) and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0)
            and m.groups == 1 and m.dilation == (1, 1) and supported(x, m.weight)
            and (m.bias is None or m.bias.dtype == torch.float32))


class Conv1x1OneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, scale=None):
        _lib.require_gpu(x, weight, scale)
        if scale is not None and (scale.dtype != torch.float32 or scale.numel() != x.shape[0] * x.shape[1]):
            raise RuntimeError("conv1x1_one: scale must hold batch * channels float32 factors")
        if not supported(x, weight) or (bias is not None and (bias.dtype != torch.float32 or bias.numel() != 1)):
            raise RuntimeError("conv1x1_one: float32 NCHW input with 16 or 64 channels and H*W % 4 == 0, a [1, C, 1, 1] "
                               "float32 weight and a float32 bias of one element required")
        x = x.contiguous()
        weight = weight.contiguous()
        B, C, H, W = x.shape
        out = torch.empty((B, 1, H, W), device=x.device, dtype=torch.float32)
        p = _lib.Conv1x1OneParams()
        p.batch, p.channels, p.hw = B, C, H * W
        p.input, p.weight, p.bias, p.out = x.data_ptr(), weight.data_ptr(), _lib.ptr(bias), out.data_ptr()
        scale = scale.contiguous() if scale is not None else None
        p.scale = _lib.ptr(scale)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_conv1x1_one_fwd(p, _lib.stream_of(x)))
        ctx.save_for_backward(x, weight, scale)
        ctx.has_bias = bias is not None
        ctx.may_defer = deferred.may_defer(weight, bias)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, scale = ctx.saved_tensors
        B, C, H, W = x.shape
        g = g.float().contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if not (need_x or need_w or need_b):
            return None, None, None, None
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(1, device=x.device, dtype=torch.float32) if need_b else None
        L = _lib.lib()
        ws = torch.empty(L.mmu_conv1x1_one_workspace_floats(B, C, H * W), device=x.device, dtype=torch.float32)
        p = _lib.Conv1x1OneParams()
        p.batch, p.channels, p.hw = B, C, H * W
        p.input, p.weight, p.dout = x.data_ptr(), weight.data_ptr(), g.data_ptr()
        p.dinput, p.dweight, p.dbias, p.workspace = _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), ws.data_ptr()
        p.scale = _lib.ptr(scale)
        with torch.cuda.device(x.device), deferred.guard(ctx.may_defer):
            _lib.check(L.mmu_conv1x1_one_bwd(p, _lib.stream_of(x)))
        deferred.keep(ws)    # (inside a deferred.Scope the sum over the workgroups' partials runs later)
        return dx, dw, db, None


def conv1x1_one(x, weight, bias=None, scale=None):
    return Conv1x1OneFn.apply(x, weight, bias, scale)


def conv_module(m, x, dropout=None):
    """``m(x)`` for an ``nn.Conv2d``: the HIP kernels when :func:`module_supported`, the module itself otherwise.
    ``dropout``: an ``nn.Dropout2d`` to apply to ``x`` first (SideoutBlock, MMUNet.py:345-350) -- its (batch, channel) mask
    is drawn exactly as the module draws it and folded into the convolution's weights per batch item instead of
    multiplied into the activation (two passes over it each way)."""
    active = dropout is not None and dropout.training and dropout.p > 0
    if not module_supported(m, x):
        return m(dropout(x) if dropout is not None else x)
    scale = None
    if active:   # F.dropout2d: noise = x.new_empty(B, C, 1, 1).bernoulli_(1 - p).div_(1 - p); x * noise
        scale = x.new_empty((x.shape[0], x.shape[1], 1, 1)).bernoulli_(1 - dropout.p).div_(1 - dropout.p)
    return conv1x1_one(x, m.weight, m.bias, scale)


# ---- nn.Conv2d(2, 1, 7, padding=3, bias=False): CBAM's spatial-attention convolution (MMUNet.py:323,335) --------------
def conv7_supported(m, x):
    return (ENABLED and isinstance(m, torch.nn.Conv2d) and m.kernel_size == (7, 7) and m.stride == (1, 1)
            and m.padding == (3, 3) and m.dilation == (1, 1) and m.groups == 1 and m.bias is None
            and m.in_channels == 2 and m.out_channels == 1 and m.padding_mode == "zeros" and x.is_cuda and x.dim() == 4
            and x.dtype == torch.float32 and m.weight.dtype == torch.float32 and not torch.is_autocast_enabled())


class Conv7x7SmallFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        _lib.require_gpu(x, weight)
        if x.dim() != 4 or x.shape[1] != 2 or tuple(weight.shape) != (1, 2, 7, 7) or x.dtype != torch.float32 \
                or weight.dtype != torch.float32:
            raise RuntimeError("conv7x7_2to1: float32 (B, 2, H, W) input and a float32 (1, 2, 7, 7) weight required")
        x, weight = x.contiguous(), weight.contiguous()
        B, _, H, W = x.shape
        out = torch.empty((B, 1, H, W), device=x.device, dtype=torch.float32)
        p = _lib.Conv7x7Params()
        p.batch, p.height, p.width = B, H, W
        p.input, p.weight, p.out = x.data_ptr(), weight.data_ptr(), out.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_conv7x7_2to1_fwd(p, _lib.stream_of(x)))
        ctx.save_for_backward(x, weight)
        ctx.may_defer = deferred.may_defer(weight)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        B, _, H, W = x.shape
        g = g.float().contiguous()
        need_x, need_w = ctx.needs_input_grad
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty_like(weight) if need_w else None
        L = _lib.lib()
        ws = torch.empty(L.mmu_conv7x7_2to1_workspace_floats(B, H, W), device=x.device, dtype=torch.float32) if need_w else None
        p = _lib.Conv7x7Params()
        p.batch, p.height, p.width = B, H, W
        p.input, p.weight, p.dout = x.data_ptr(), weight.data_ptr(), g.data_ptr()
        p.dinput, p.dweight, p.workspace = _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(ws)
        with torch.cuda.device(x.device), deferred.guard(ctx.may_defer):
            _lib.check(L.mmu_conv7x7_2to1_bwd(p, _lib.stream_of(x)))
        deferred.keep(ws)    # (inside a deferred.Scope the sum over the blocks' partials runs later)
        return dx, dw


def conv7_module(m, x):
    """``m(x)`` for CBAM's 7 x 7 convolution: the HIP kernels when :func:`conv7_supported`, the module otherwise."""
    return Conv7x7SmallFn.apply(x, m.weight) if conv7_supported(m, x) else m(x)


def _need_f32_nchw(name, t, hw_mult=4):
    """Raw-pointer rule: a wrapper takes ``data_ptr()`` only of tensors it has itself checked to be
    float32 with the expected rank; anything else raises instead of being read as float32 out of bounds."""
    if t.dtype != torch.float32 or t.dim() != 4 or (t.shape[2] * t.shape[3]) % hw_mult != 0 or t.shape[0] >= 65536:
        raise RuntimeError(f"{name}: expected a float32 (B, C, H, W) tensor with H*W % {hw_mult} == 0 and B < 65536, got "
                           f"{t.dtype} {tuple(t.shape)}")


# ---- x * gate with a per-channel ([B, C, 1, 1]) or per-pixel ([B, 1, H, W]) gate (MMUNet.py:330,336,415) --------------
GATED_MUL = os.environ.get("MMUNET_GATED_MUL", "1") != "0"


def _gate_mode(x, gate):
    if not (ENABLED and GATED_MUL and x.is_cuda and x.dim() == 4 and gate.dim() == 4 and x.dtype == torch.float32
            and gate.dtype == torch.float32 and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[0] < 65536
            and not torch.is_autocast_enabled()):
        return None
    B, C, H, W = x.shape
    if tuple(gate.shape) == (B, C, 1, 1):
        return 0
    if tuple(gate.shape) == (B, 1, H, W) and C > 1:
        return 1
    return None


def _spatial_workspace(x, mode):
    """Chunk sums of the spatial gate's gradient (csrc/gated_mul.hip), or None when one chunk covers the channels."""
    n = _lib.lib().mmu_gated_mul_bwd_workspace_floats(x.shape[0], x.shape[1], x.shape[2] * x.shape[3], mode)
    return torch.empty(n, device=x.device, dtype=torch.float32) if n else None


class ChannelStatsSlot:
    """Hand-over between ``channel_max_mean(y)`` and the ``gated_mul`` that produced ``y`` (CBAM: y1 = x * c_out feeds the
    spatial statistics and the second product): the statistics' backward parks its two small operands here and returns
    no gradient; the product's backward adds the gradient they stand for to its ``dout`` on the fly (csrc/gated_mul.hip)
    -- the 134 MB statistics gradient of the stem map is neither written nor added."""
    __slots__ = ("armed", "parked")

    def __init__(self):
        self.armed, self.parked = False, None


class GatedMulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate, mode, x_slot=None, stats_slot=None):
        _lib.require_gpu(x, gate)
        _need_f32_nchw("gated_mul", x)
        if gate.dtype != torch.float32 or mode not in (0, 1) or tuple(gate.shape) != (
                (x.shape[0], x.shape[1], 1, 1) if mode == 0 else (x.shape[0], 1, x.shape[2], x.shape[3])):
            raise RuntimeError(f"gated_mul: gate must be float32 (B, C, 1, 1) [mode 0] or (B, 1, H, W) [mode 1], got "
                               f"{gate.dtype} {tuple(gate.shape)} for mode {mode} and input {tuple(x.shape)}")
        x, gate = x.contiguous(), gate.contiguous()
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        p = _lib.GatedMulParams()
        p.batch, p.channels, p.mode, p.hw = B, C, mode, H * W
        p.input, p.gate, p.out = x.data_ptr(), gate.data_ptr(), out.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_gated_mul_fwd(p, _lib.stream_of(x)))
        ctx.save_for_backward(x, gate)
        ctx.mode = mode
        # x_slot: a conv3x3_small.SharedGrad of the consumers of x; stats_slot: see ChannelStatsSlot
        ctx.x_slot = x_slot if (x_slot is not None and ctx.needs_input_grad[0]) else None
        if ctx.x_slot is not None:
            ctx.x_slot.join()
        ctx.stats_slot = stats_slot if mode == 0 else None
        if ctx.stats_slot is not None:
            ctx.stats_slot.armed = True
        return out

    @staticmethod
    def backward(ctx, g):
        x, gate = ctx.saved_tensors
        B, C, H, W = x.shape
        g = g.float().contiguous()
        need_x, need_g = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        stats = None
        if ctx.stats_slot is not None and ctx.stats_slot.parked is not None:
            stats, ctx.stats_slot.parked = ctx.stats_slot.parked, None
        if not (need_x or need_g):
            return None, None, None, None, None
        dx = torch.empty_like(x) if need_x else None
        dgate = torch.empty_like(gate) if need_g else None
        p = _lib.GatedMulParams()
        p.batch, p.channels, p.mode, p.hw = B, C, ctx.mode, H * W
        p.input, p.gate, p.dout = x.data_ptr(), gate.data_ptr(), g.data_ptr()
        p.dinput, p.dgate = _lib.ptr(dx), _lib.ptr(dgate)
        ws = _spatial_workspace(x, ctx.mode) if need_g else None
        p.workspace = _lib.ptr(ws)
        if stats is not None:
            sg, sam = stats
            if sg.shape != (B, 2, H, W) or sam.shape != (B, H, W) or sg.dtype != torch.float32 or sam.dtype != torch.int32:
                raise RuntimeError("gated_mul: parked channel statistics do not match the product")
            p.stats_dout, p.stats_argmax = sg.data_ptr(), sam.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_gated_mul_bwd(p, _lib.stream_of(x)))
        if ctx.x_slot is not None:
            parked = ctx.x_slot.take()
            dx = ctx.x_slot.give(dx if parked is None else parked.add_(dx))
        return dx, dgate, None, None, None


class GatedMul3Fn(torch.autograd.Function):
    """``x * x2 * gate + addend`` with a per-pixel gate (B, 1, H, W): RCG's ``x0 * gate * x2 + f`` (MMUNet.py:415) in one pass
    each way instead of a product, a gated product and an add (three passes) / five kernels backward."""

    @staticmethod
    def forward(ctx, x, x2, gate, addend):
        _lib.require_gpu(x, x2, gate, addend)
        _need_f32_nchw("gated_mul3", x)
        B, C, H, W = x.shape
        if any(t.dtype != torch.float32 for t in (x2, gate, addend)) or x2.shape != x.shape or addend.shape != x.shape or \
                tuple(gate.shape) != (B, 1, H, W):
            raise RuntimeError("gated_mul3: float32 x, x2, addend of one (B, C, H, W) shape and a (B, 1, H, W) gate required")
        x, x2, gate, addend = x.contiguous(), x2.contiguous(), gate.contiguous(), addend.contiguous()
        out = torch.empty_like(x)
        p = _lib.GatedMulParams()
        p.batch, p.channels, p.mode, p.hw = B, C, 1, H * W
        p.input, p.gate, p.out, p.input2, p.addend = x.data_ptr(), gate.data_ptr(), out.data_ptr(), x2.data_ptr(), addend.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_gated_mul_fwd(p, _lib.stream_of(x)))
        ctx.save_for_backward(x, x2, gate)
        return out

    @staticmethod
    def backward(ctx, g):
        x, x2, gate = ctx.saved_tensors
        B, C, H, W = x.shape
        g = g.float().contiguous()
        need = ctx.needs_input_grad
        dx = torch.empty_like(x) if need[0] else None
        dx2 = torch.empty_like(x) if need[1] else None
        dgate = torch.empty_like(gate) if need[2] else None
        if need[0] or need[1] or need[2]:
            p = _lib.GatedMulParams()
            p.batch, p.channels, p.mode, p.hw = B, C, 1, H * W
            p.input, p.gate, p.dout, p.input2 = x.data_ptr(), gate.data_ptr(), g.data_ptr(), x2.data_ptr()
            p.dinput, p.dgate, p.dinput2 = _lib.ptr(dx), _lib.ptr(dgate), _lib.ptr(dx2)
            ws = _spatial_workspace(x, 1) if need[2] else None
            p.workspace = _lib.ptr(ws)
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().mmu_gated_mul_bwd(p, _lib.stream_of(x)))
        return dx, dx2, dgate, (g if need[3] else None)


def gated_mul3(x, x2, gate, addend):
    """``x * x2 * gate + addend`` for a per-pixel gate; the one-pass HIP form for float32 NCHW tensors, else ATen."""
    if (_gate_mode(x, gate) == 1 and x2.shape == x.shape and addend.shape == x.shape and x2.dtype == torch.float32
            and addend.dtype == torch.float32):
        return GatedMul3Fn.apply(x, x2, gate, addend)
    return gated_mul(x * x2, gate) + addend


def gated_mul(x, gate, x_slot=None, stats_slot=None):
    """``x * gate`` (broadcast); the one-pass HIP form for channel / spatial gates of a float32 NCHW tensor, else ATen.
    ``x_slot``: a conv3x3_small.SharedGrad of the consumers of ``x``; ``stats_slot``: a ChannelStatsSlot shared with the
    ``channel_max_mean`` of the product."""
    mode = _gate_mode(x, gate)
    return GatedMulFn.apply(x, gate, mode, x_slot, stats_slot) if mode is not None else x * gate


# ---- CBAM's pooled statistics (MMUNet.py:327-333): mean and max in one pass, one-pass backward ---------------------------
STATS = os.environ.get("MMUNET_CBAM_STATS", "1") != "0"


def stats_supported(x):
    return (ENABLED and STATS and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32
            and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[0] < 65536 and x.shape[1] > 1
            and not torch.is_autocast_enabled())


class _PixelStatsFn(torch.autograd.Function):
    """x (B, C, H, W) -> (mean, max) over the pixels, each (B, C, 1, 1)."""

    @staticmethod
    def forward(ctx, x, x_slot=None):
        _lib.require_gpu(x)
        _need_f32_nchw("pixel_mean_max", x)
        x = x.contiguous()
        B, C, H, W = x.shape
        mean = torch.empty((B, C, 1, 1), device=x.device, dtype=torch.float32)
        mx = torch.empty_like(mean)
        am = torch.empty((B, C), device=x.device, dtype=torch.int32)
        p = _lib.CbamStatsParams()
        p.batch, p.channels, p.mode, p.hw = B, C, 0, H * W
        p.input, p.mean, p.max, p.argmax = x.data_ptr(), mean.data_ptr(), mx.data_ptr(), am.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_cbam_stats_fwd(p, _lib.stream_of(x)))
        ctx.save_for_backward(am)
        ctx.shape = x.shape
        ctx.x_slot = x_slot if (x_slot is not None and ctx.needs_input_grad[0]) else None
        if ctx.x_slot is not None:
            ctx.x_slot.join()
        return mean, mx

    @staticmethod
    def backward(ctx, gmean, gmax):
        am, = ctx.saved_tensors
        B, C, H, W = ctx.shape
        gmean, gmax = gmean.float().contiguous(), gmax.float().contiguous()
        parked = ctx.x_slot.take() if ctx.x_slot is not None else None
        if parked is not None and (parked.shape != ctx.shape or parked.dtype != torch.float32 or not parked.is_contiguous()):
            raise RuntimeError("pixel_mean_max: parked input gradient does not match the input")
        dx = parked if parked is not None else torch.empty(ctx.shape, device=am.device, dtype=torch.float32)   # (in place)
        p = _lib.CbamStatsParams()
        p.batch, p.channels, p.mode, p.hw = B, C, 0, H * W
        p.argmax, p.dmean, p.dmax, p.dinput = am.data_ptr(), gmean.data_ptr(), gmax.data_ptr(), dx.data_ptr()
        p.dinput_addend = _lib.ptr(parked)
        with torch.cuda.device(am.device):
            _lib.check(_lib.lib().mmu_cbam_stats_bwd(p, _lib.stream_of(am)))
        return (ctx.x_slot.give(dx) if ctx.x_slot is not None else dx), None


class _ChannelStatsFn(torch.autograd.Function):
    """x (B, C, H, W) -> (B, 2, H, W): max over the channels, then their mean (the order of MMUNet.py:333)."""

    @staticmethod
    def forward(ctx, x, stats_slot=None):
        _lib.require_gpu(x)
        _need_f32_nchw("channel_max_mean", x)
        x = x.contiguous()
        B, C, H, W = x.shape
        out = torch.empty((B, 2, H, W), device=x.device, dtype=torch.float32)
        am = torch.empty((B, H, W), device=x.device, dtype=torch.int32)
        p = _lib.CbamStatsParams()
        p.batch, p.channels, p.mode, p.hw = B, C, 1, H * W
        p.input, p.out, p.argmax = x.data_ptr(), out.data_ptr(), am.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_cbam_stats_fwd(p, _lib.stream_of(x)))
        ctx.save_for_backward(am)
        ctx.shape = x.shape
        ctx.stats_slot = stats_slot if (stats_slot is not None and stats_slot.armed and ctx.needs_input_grad[0]) else None
        return out

    @staticmethod
    def backward(ctx, g):
        am, = ctx.saved_tensors
        B, C, H, W = ctx.shape
        g = g.float().contiguous()
        if ctx.stats_slot is not None:       # the product that made x adds this gradient itself (ChannelStatsSlot)
            ctx.stats_slot.parked = (g, am)
            return None, None
        dx = torch.empty(ctx.shape, device=am.device, dtype=torch.float32)
        p = _lib.CbamStatsParams()
        p.batch, p.channels, p.mode, p.hw = B, C, 1, H * W
        p.argmax, p.dout, p.dinput = am.data_ptr(), g.data_ptr(), dx.data_ptr()
        with torch.cuda.device(am.device):
            _lib.check(_lib.lib().mmu_cbam_stats_bwd(p, _lib.stream_of(am)))
        return dx, None


class _CbamGateFn(torch.autograd.Function):
    """(avg, max) (B, C, 1, 1), w1 (R, C, 1, 1), w2 (C, R, 1, 1) -> sigmoid(mlp(avg) + mlp(max)) (B, C, 1, 1)."""

    @staticmethod
    def forward(ctx, avg, mx, w1, w2):
        _lib.require_gpu(avg, mx, w1, w2)
        B, C = avg.shape[:2]
        R = w1.shape[0]
        ts = (avg, mx, w1, w2)
        if any(t.dtype != torch.float32 for t in ts) or tuple(avg.shape) != (B, C, 1, 1) or mx.shape != avg.shape or \
                tuple(w1.shape) != (R, C, 1, 1) or tuple(w2.shape) != (C, R, 1, 1):
            raise RuntimeError("cbam_gate: float32 (B, C, 1, 1) vectors with (R, C, 1, 1) / (C, R, 1, 1) weights required")
        avg, mx, w1, w2 = (t.contiguous() for t in ts)
        gate = torch.empty_like(avg)
        p = _lib.CbamGateParams()
        p.batch, p.channels, p.hidden = B, C, R
        p.avg, p.max, p.w1, p.w2, p.gate = avg.data_ptr(), mx.data_ptr(), w1.data_ptr(), w2.data_ptr(), gate.data_ptr()
        with torch.cuda.device(avg.device):
            _lib.check(_lib.lib().mmu_cbam_gate_fwd(p, _lib.stream_of(avg)))
        ctx.save_for_backward(avg, mx, w1, w2, gate)
        return gate

    @staticmethod
    def backward(ctx, g):
        avg, mx, w1, w2, gate = ctx.saved_tensors
        B, C = avg.shape[:2]
        g = g.float().contiguous()
        need = ctx.needs_input_grad
        davg = torch.empty_like(avg) if need[0] else None
        dmax = torch.empty_like(mx) if need[1] else None
        dw1 = torch.empty_like(w1) if need[2] else None
        dw2 = torch.empty_like(w2) if need[3] else None
        p = _lib.CbamGateParams()
        p.batch, p.channels, p.hidden = B, C, w1.shape[0]
        p.avg, p.max, p.w1, p.w2, p.gate = avg.data_ptr(), mx.data_ptr(), w1.data_ptr(), w2.data_ptr(), gate.data_ptr()
        p.dgate, p.davg, p.dmax, p.dw1, p.dw2 = g.data_ptr(), _lib.ptr(davg), _lib.ptr(dmax), _lib.ptr(dw1), _lib.ptr(dw2)
        with torch.cuda.device(avg.device):
            _lib.check(_lib.lib().mmu_cbam_gate_bwd(p, _lib.stream_of(avg)))
        return davg, dmax, dw1, dw2


CBAM_GATE = os.environ.get("MMUNET_CBAM_GATE", "1") != "0"


def cbam_gate_supported(mlp, v):
    """CBAM's shared MLP as the reference builds it (two bias-free 1 x 1 convolutions around a ReLU) on float32 pooled
    vectors small enough for one workgroup."""
    import torch.nn as nn
    if not (ENABLED and CBAM_GATE and v.is_cuda and v.dtype == torch.float32 and v.dim() == 4 and v.shape[2:] == (1, 1)
            and not torch.is_autocast_enabled() and isinstance(mlp, nn.Sequential) and len(mlp) == 3):
        return False
    a, r, b = mlp
    ok = lambda c: (isinstance(c, nn.Conv2d) and c.kernel_size == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0)   # noqa: E731
                    and c.groups == 1 and c.bias is None and c.weight.dtype == torch.float32)
    return (ok(a) and ok(b) and isinstance(r, nn.ReLU) and a.in_channels == v.shape[1] and b.out_channels == v.shape[1]
            and a.out_channels == b.in_channels and v.shape[0] * (v.shape[1] + 4 * a.out_channels) * 4 <= 48 * 1024)


def cbam_gate(mlp, avg, mx):
    """``sigmoid(mlp(avg) + mlp(mx))`` (MMUNet.py:329) in one launch each way."""
    return _CbamGateFn.apply(avg, mx, mlp[0].weight, mlp[2].weight)


def pixel_mean_max(x, x_slot=None):
    """(avg_pool(x), max_pool(x)) of CBAM's channel attention, each (B, C, 1, 1).  ``x_slot``: a conv3x3_small.SharedGrad
    of the consumers of ``x``."""
    return _PixelStatsFn.apply(x, x_slot)


def channel_max_mean(x, stats_slot=None):
    """cat((max over channels, mean over channels), 1) of CBAM's spatial attention, (B, 2, H, W).  ``stats_slot``: the
    ChannelStatsSlot the ``gated_mul`` that produced ``x`` was given."""
    return _ChannelStatsFn.apply(x, stats_slot)
