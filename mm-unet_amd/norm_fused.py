"""``gn_bn_act`` -- GroupNorm [-> BatchNorm2d] [-> ReLU | tanh] as one normalisation (HIP, fwd + bwd).

Every MMConv ends in GroupNorm (MMUNet.py:265) and is followed by BatchNorm2d and usually ReLU in the
blocks that use it (:344-349, 357-359, 424-430, 436-452); its offset branch is GroupNorm -> tanh (:250).
Both normalisations are affine in x per (batch, channel) once their statistics are known,

    GroupNorm   y1 = a x + d          a = gamma_g r_bg,  d = beta_g - a mu_bg
    BatchNorm   y2 = A x + D          mean_c(y1) = 1/N sum_b (a s1 + d HW),  E_c(y1^2) = 1/N sum_b (a^2 s2 + 2 a d s1 + d^2 HW)

with s1 = sum_hw x, s2 = sum_hw x^2 per (batch, channel): ONE moments pass gives both sets of statistics and
one apply pass writes act(A x + D).  Backward, with g2 = dout * act'(y2), t1 = sum_hw g2, t2 = sum_hw g2 x:

    BatchNorm (training)   g1 = k (g2 - e - (p x + q) f),   k = gamma_b r_b, e = dbeta_b / N, f = dgamma_b / N,
                           p = a r_b, q = (d - m_c) r_b,  dbeta_b = sum_b t1,  dgamma_b = sum_b (p t2 + q t1)
    per (b, c)             u1 = sum_hw g1 = k (t1 - e HW - f (p s1 + q HW)),  u2 = sum_hw g1 x = k (t2 - e s1 - f (p s2 + q s1))
    GroupNorm              M1 = mean_g(gamma_g g1),  M2 = mean_g(gamma_g g1 xhat)   from u1, u2
                           dx = r [gamma_g g1 - M1 - xhat M2]  =  c0 g2 + c1 x + c2    (per-(b, c) constants)

so the backward is one pass for (t1, t2) and one for dx.  As separate ATen / MIOpen ops the chain makes 8
passes over the activation forward and 13 backward.  NCHW contiguous; the input is float32 or bfloat16 (what its
producer emits: the tokens-last GEMM float32, a library convolution under autocast bfloat16), the output is
bfloat16 under bf16 autocast (and for a bfloat16 input) and float32 otherwise; statistics, parameters and all
arithmetic are float32 (group_norm / batch_norm are on autocast's float32 list: the reference computes them in
float32 there as well and rounds the result once).
"""
import torch
import torch.nn as nn

from . import _lib

ACT = {None: 0, "none": 0, "relu": 1, "tanh": 2}
_IO = (torch.float32, torch.bfloat16)


def _out_dtype(x):
    """bfloat16 under bf16 autocast or for a bfloat16 input, float32 otherwise."""
    if x.dtype == torch.bfloat16:
        return torch.bfloat16
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return torch.bfloat16
    return torch.float32


class GnBnActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gn_w, gn_b, bn_w, bn_b, pre_bias, run_mean, run_var, groups, gn_eps, has_bn, training, bn_eps,
                momentum, act, grad_cb=False, residual=None, out_dtype=None):
        _lib.require_gpu(x)
        if x.dtype not in _IO:
            raise RuntimeError("gn_bn_act: float32 or bfloat16 input required")
        x = x.contiguous()
        B, C, H, W = x.shape
        dev, f32 = x.device, torch.float32
        act_dtype = out_dtype if out_dtype is not None else _out_dtype(x)
        out = torch.empty(x.shape, device=dev, dtype=act_dtype)
        stats = torch.empty(4 * B * C + 2 * B * groups + 2 * C, device=dev, dtype=f32)
        s1, s2, scale, shift = (stats[i * B * C:(i + 1) * B * C] for i in range(4))
        mu = stats[4 * B * C:4 * B * C + B * groups]
        rstd = stats[4 * B * C + B * groups:4 * B * C + 2 * B * groups]
        bn_mean = stats[4 * B * C + 2 * B * groups:4 * B * C + 2 * B * groups + C]
        bn_rstd = stats[4 * B * C + 2 * B * groups + C:]
        p = _lib.NormParams()
        p.batch, p.channels, p.groups, p.hw = B, C, groups, H * W
        p.has_bn, p.training, p.act, p.has_gn = int(has_bn), int(training), act, int(gn_eps >= 0)
        p.gn_eps, p.bn_eps, p.momentum = max(gn_eps, 0.0), bn_eps, momentum
        p.input, p.out = x.data_ptr(), out.data_ptr()
        p.x_dtype, p.act_dtype = _lib.dtype_code(x), _lib.dtype_code(out)
        p.gn_weight, p.gn_bias, p.bn_weight, p.bn_bias = (_lib.ptr(t) for t in (gn_w, gn_b, bn_w, bn_b))
        p.pre_bias = _lib.ptr(pre_bias)
        ctx.res_dtype = None
        if residual is not None:
            if residual.shape != x.shape or residual.device != x.device:
                raise RuntimeError("gn_bn_act: residual must have the shape / device of the input")
            ctx.res_dtype = residual.dtype
            residual = residual.to(act_dtype).contiguous()   # (read in the output's type; the sum itself is fp32)
            p.residual = residual.data_ptr()
        for name, t in (("gn weight", gn_w), ("gn bias", gn_b), ("bn weight", bn_w), ("bn bias", bn_b),
                        ("pre_bias", pre_bias), ("running_mean", run_mean), ("running_var", run_var)):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != C
                                  or t.device != dev):
                raise RuntimeError(f"gn_bn_act: {name} must be a contiguous float32 vector of {C} elements on {dev}")
        if C % groups != 0:
            raise RuntimeError("gn_bn_act: channels must be divisible by groups")
        p.running_mean, p.running_var = _lib.ptr(run_mean), _lib.ptr(run_var)
        p.s1, p.s2, p.mu, p.rstd = s1.data_ptr(), s2.data_ptr(), mu.data_ptr(), rstd.data_ptr()
        p.bn_mean, p.bn_rstd, p.scale, p.shift = bn_mean.data_ptr(), bn_rstd.data_ptr(), scale.data_ptr(), shift.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mmu_norm_fused_fwd(p, _lib.stream_of(x)))
        ctx.has_res = residual is not None
        ctx.act_dtype = act_dtype
        ctx.save_for_backward(x, gn_w, gn_b, bn_w, bn_b, pre_bias, stats, out if ctx.has_res else None)
        ctx.cfg = (groups, gn_eps, has_bn, training, bn_eps, momentum, act)
        ctx.grad_cb = bool(grad_cb)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gn_w, gn_b, bn_w, bn_b, pre_bias, stats, out_saved = ctx.saved_tensors
        groups, gn_eps, has_bn, training, bn_eps, momentum, act = ctx.cfg
        B, C, H, W = x.shape
        dev, f32 = x.device, torch.float32
        act_dtype = ctx.act_dtype
        g = dout.to(act_dtype).contiguous()
        if g.shape != x.shape:
            raise RuntimeError("gn_bn_act backward: gradient shape mismatch")
        if ctx.grad_cb:   # the producer of x is a tokens-last GEMM: hand it its gradient as [C][B][HW] (no copy there)
            dx = torch.empty((C, B, H, W), device=dev, dtype=x.dtype).permute(1, 0, 2, 3)
        else:
            dx = torch.empty_like(x)
        L = _lib.lib()
        ws = torch.empty(L.mmu_norm_fused_workspace_floats(B, C, groups), device=dev, dtype=f32)
        grads = torch.empty(5 * C, device=dev, dtype=f32)
        s1, s2, scale, shift = (stats[i * B * C:(i + 1) * B * C] for i in range(4))
        mu = stats[4 * B * C:4 * B * C + B * groups]
        rstd = stats[4 * B * C + B * groups:4 * B * C + 2 * B * groups]
        bn_mean = stats[4 * B * C + 2 * B * groups:4 * B * C + 2 * B * groups + C]
        bn_rstd = stats[4 * B * C + 2 * B * groups + C:]
        p = _lib.NormParams()
        p.batch, p.channels, p.groups, p.hw = B, C, groups, H * W
        p.has_bn, p.training, p.act, p.has_gn = int(has_bn), int(training), act, int(gn_eps >= 0)
        p.gn_eps, p.bn_eps, p.momentum = max(gn_eps, 0.0), bn_eps, momentum
        p.input, p.dout, p.dinput = x.data_ptr(), g.data_ptr(), dx.data_ptr()
        p.x_dtype, p.act_dtype = _lib.dtype_code(x), _lib.dtype_code(g)
        p.dinput_channel_major = int(ctx.grad_cb)
        dres = None
        if ctx.has_res:
            dres = torch.empty(x.shape, device=dev, dtype=act_dtype)
            p.act_out, p.dresidual = out_saved.data_ptr(), dres.data_ptr()
        p.gn_weight, p.gn_bias, p.bn_weight, p.bn_bias = (_lib.ptr(t) for t in (gn_w, gn_b, bn_w, bn_b))
        p.pre_bias = _lib.ptr(pre_bias)
        p.s1, p.s2, p.mu, p.rstd = s1.data_ptr(), s2.data_ptr(), mu.data_ptr(), rstd.data_ptr()
        p.bn_mean, p.bn_rstd, p.scale, p.shift = bn_mean.data_ptr(), bn_rstd.data_ptr(), scale.data_ptr(), shift.data_ptr()
        dgw, dgb, dbw, dbb, dpb = (grads[i * C:(i + 1) * C] for i in range(5))
        if gn_eps < 0:   # BatchNorm alone: no GroupNorm parameters
            dgw = dgb = None
        if pre_bias is not None:
            p.dpre_bias = dpb.data_ptr()
        p.dgn_weight, p.dgn_bias = _lib.ptr(dgw), _lib.ptr(dgb)
        if has_bn:
            p.dbn_weight, p.dbn_bias = dbw.data_ptr(), dbb.data_ptr()
        p.workspace = ws.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(L.mmu_norm_fused_bwd(p, _lib.stream_of(x)))
        return (dx, dgw if gn_w is not None else None, dgb if gn_b is not None else None,
                dbw if (has_bn and bn_w is not None) else None, dbb if (has_bn and bn_b is not None) else None,
                dpb if pre_bias is not None else None, None, None, None, None, None, None, None, None, None, None,
                dres if dres is None else dres.to(ctx.res_dtype), None)


class batched_counters:
    """Context: the ``num_batches_tracked += 1`` of every BatchNorm2d that runs through this module inside it is
    collected and applied as ONE multi-tensor add on exit (56 one-element launches per MM_Net training step
    otherwise).  Same values as the modules' own increments; nothing reads the counters in between (momentum is a
    number everywhere in MM-UNet)."""
    _active = None

    def __enter__(self):
        self._prev, batched_counters._active = batched_counters._active, []
        return self

    def __exit__(self, *exc):
        pending, batched_counters._active = batched_counters._active, self._prev
        if pending and exc[0] is None:
            torch._foreach_add_(pending, 1)
        return False


def _count_batch(bn):
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if batched_counters._active is not None:
            batched_counters._active.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)


ENABLED = True   # False: GroupNorm / BatchNorm2d / activation stay separate module calls (fused_paths.plain_aten)


def supported(x, gn, bn=None):
    ok = ENABLED and x.is_cuda and x.dtype in _IO and x.dim() == 4 and \
        x.shape[0] * x.shape[1] < 65536 and isinstance(gn, nn.GroupNorm)
    if bn is not None:
        ok = ok and isinstance(bn, nn.BatchNorm2d) and (bn.track_running_stats or bn.training) and \
            bn.momentum is not None
    return ok


def bn_act_supported(x, bn):
    return ENABLED and x.is_cuda and x.dtype in _IO and x.dim() == 4 and \
        x.shape[0] * x.shape[1] < 65536 and isinstance(bn, nn.BatchNorm2d) and bn.affine and \
        (bn.track_running_stats or bn.training) and bn.momentum is not None


def bn_act(x, bn, act=None, residual=None, pre_bias=None, out_dtype=None):
    """``act(bn(x + pre_bias[None, :, None, None]))`` for an ``nn.BatchNorm2d`` (training or eval statistics): the
    same two-pass kernels with the GroupNorm stage switched off -- BatchNorm and ReLU read and write the
    activation once each way together instead of once each.  ``pre_bias``: the bias of the convolution that
    produced ``x``, folded into the statistics (its gradient comes out of the same backward algebra, so the
    convolution needs neither a bias add nor a bias-gradient reduction over the whole activation)."""
    training = bool(bn.training or not bn.track_running_stats)
    _count_batch(bn)
    return GnBnActFn.apply(x, None, None, bn.weight, bn.bias, pre_bias, bn.running_mean, bn.running_var, x.shape[1],
                           -1.0, True, training, bn.eps, bn.momentum, ACT[act], False, residual, out_dtype)


def gn_bn_act(x, gn, bn=None, act=None, pre_bias=None, grad_channel_major=False, residual=None, out_dtype=None):
    """``act(bn(gn(x + pre_bias[None, :, None, None])))`` with ``gn`` an ``nn.GroupNorm``, ``bn`` an optional
    ``nn.BatchNorm2d`` (its running statistics are updated in training mode exactly as the module would),
    ``act`` in {None, "relu", "tanh"}; ``pre_bias`` (the bias of the convolution that produced ``x``) is folded
    into the statistics instead of being added to the activation.  ``grad_channel_major``: return d x laid out
    [C][B][HW] (what a tokens-last GEMM producer of ``x`` wants; saves it a transposing copy).  ``residual``
    (needs ``act="relu"``): ``relu(bn(gn(x)) + residual)`` -- the tail of a ResidualBlock (MMUNet.py:455-467) in
    the same two passes; its gradient ``dout * (out > 0)`` is written by the backward apply pass.  ``out_dtype``:
    float32 / bfloat16 (default: bfloat16 under bf16 autocast or for a bfloat16 input, else float32)."""
    if residual is not None and act != "relu":
        raise ValueError("gn_bn_act: a residual input needs act='relu'")
    has_bn = bn is not None
    training = bool(has_bn and (bn.training or not bn.track_running_stats))
    if has_bn:
        _count_batch(bn)
    return GnBnActFn.apply(x, gn.weight, gn.bias, bn.weight if has_bn else None, bn.bias if has_bn else None,
                           pre_bias, bn.running_mean if has_bn else None, bn.running_var if has_bn else None, gn.num_groups,
                           gn.eps, has_bn, training, bn.eps if has_bn else 0.0, bn.momentum if has_bn else 0.0, ACT[act],
                           grad_channel_major, residual, out_dtype)
