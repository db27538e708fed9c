"""MMConv's K x 1 DSC convolution with the channel mixing BEFORE the deformable sampling (csrc/morph_mix.hip).

``dsc_conv_x(get_interpolated_feature(x))`` (src/UM_Net/MMUNet.py:259-263) is
``out[o] = sum_c sum_k W[o, c, k] S_k(x[c])`` with S_k the bilinear row sampling of tap k -- linear, and the same for every
channel.  So ``out[o] = sum_k S_k(Y[k * O + o])`` with ``Y = Wr @ x`` a 1 x 1 convolution (``Wr[k * O + o, c] = W[o, c, k]``):
the tensor that goes through HBM between the two steps has K * Cout planes instead of K * Cin.  Used by MMConv for the
blocks that reduce the channel count on large maps (DecoderBlock.conv1, SideoutBlock.conv1, RCG.conv1:
MMUNet.py:344-349,357-359,424-430); everything else keeps sample-then-mix (morph_sample + dsc_gemm).
float32, taps in (1, 3); no CPU path.
"""
import os

import torch

from . import _lib
from .tall_gemm import proj_bcl

ENABLED = os.environ.get("MMUNET_MORPH_MIX", "1") != "0"   # False: every block samples first (A/B runs, plain_aten)
MIN_PIXELS = int(os.environ.get("MMUNET_MORPH_MIX_MIN_PIXELS", "16384"))   # per map: below, the launches are floor-bound
RATIO = float(os.environ.get("MMUNET_MORPH_MIX_RATIO", "2"))                # mix first when Cin >= RATIO * Cout


def wanted(x, conv, K):
    """Mix first when it shrinks the intermediate at least 2 x, on maps large enough to be bandwidth-bound."""
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and K in (1, 3) and not torch.is_autocast_enabled()
            and RATIO * conv.out_channels <= conv.in_channels and x.shape[2] * x.shape[3] >= MIN_PIXELS
            and conv.weight.dtype == torch.float32)


class _ParkGradFn(torch.autograd.Function):
    """Identity on the block's input in front of the mixing GEMM: its gradient is parked in the conv3x3_small.GradSlot
    the offset convolution adds its own input gradient to (see there) instead of going back to autograd as a second
    gradient of the same tensor."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        slot = ctx.slot
        if slot is not None and slot.armed and slot.grad is None and g.is_contiguous() and g.dtype == torch.float32:
            slot.grad = g
            return None, None
        return g, None


class MixSampleFn(torch.autograd.Function):
    """``out[b, o, h, w] = sum_k lerp(Y[b, k*O + o, y0_k, col_k], Y[b, k*O + o, y0_k + 1, col_k], wy_k)``."""

    @staticmethod
    def forward(ctx, mixed, y, out_channels):
        _lib.require_gpu(mixed, y)
        B, KO, H, W = mixed.shape
        K = y.shape[1]
        # (mixed may carry padding planes behind the K * O real ones: the GEMM that made it wants a multiple of 64 rows)
        if mixed.dtype != torch.float32 or y.dtype != torch.float32 or KO < K * out_channels or K not in (1, 3) \
                or tuple(y.shape) != (B, K, H, W):
            raise RuntimeError("morph_mix: float32 mixed (B, >= K*O, H, W) and y (B, K, H, W) with K in (1, 3) required")
        if mixed.stride(3) != 1 or mixed.stride(2) != W:
            mixed = mixed.contiguous()
        y = y.contiguous()
        out = torch.empty((B, out_channels, H, W), device=mixed.device, dtype=torch.float32)
        p = _lib.MorphMixParams()
        p.batch, p.out_channels, p.height, p.width, p.taps = B, out_channels, H, W, K
        p.mixed, p.mixed_bs, p.mixed_cs = mixed.data_ptr(), mixed.stride(0), mixed.stride(1)
        p.y, p.out = y.data_ptr(), out.data_ptr()
        with torch.cuda.device(mixed.device):
            _lib.check(_lib.lib().mmu_morph_mix_sample_fwd(p, _lib.stream_of(mixed)))
        ctx.save_for_backward(mixed, y)
        ctx.O = out_channels
        return out

    @staticmethod
    def backward(ctx, dout):
        mixed, y = ctx.saved_tensors
        B, KO, H, W = mixed.shape
        K = y.shape[1]
        g = dout.float().contiguous()
        dmixed = torch.empty_strided(mixed.shape, mixed.stride(), device=mixed.device, dtype=torch.float32)
        if KO > K * ctx.O:
            dmixed[:, K * ctx.O:].zero_()        # padding planes: no gradient (the kernel writes the real planes only)
        dy = torch.empty_like(y)
        p = _lib.MorphMixParams()
        p.batch, p.out_channels, p.height, p.width, p.taps = B, ctx.O, H, W, K
        p.mixed, p.mixed_bs, p.mixed_cs = mixed.data_ptr(), mixed.stride(0), mixed.stride(1)
        p.y, p.dout, p.dmixed, p.dy = y.data_ptr(), g.data_ptr(), dmixed.data_ptr(), dy.data_ptr()
        with torch.cuda.device(mixed.device):
            _lib.check(_lib.lib().mmu_morph_mix_sample_bwd(p, _lib.stream_of(mixed)))
        return dmixed, dy, None


def mix_sample(mixed, y, out_channels):
    return MixSampleFn.apply(mixed, y, out_channels)


def dsc_mix_first(x, y_rows, conv, slot=None):
    """The K x 1 DSC convolution ``conv`` of the deformable samples of ``x`` at rows ``y_rows`` (B, K, H, W), WITHOUT the
    bias (the caller folds it into the normalisation): channel mixing as one strided-batch GEMM over the pixels, then
    sampling + the sum over the taps."""
    B, C, H, W = x.shape
    O, K = conv.out_channels, y_rows.shape[1]
    wr = conv.weight.reshape(O, C, K).permute(2, 0, 1).reshape(K * O, C)        # Wr[k*O + o, c] = W[o, c, k, 0]
    pad = (-K * O) % 64
    if pad and C % 16 == 0:      # a multiple of 64 rows: ONE matrix-core GEMM instead of a library call per batch item
        wr = torch.nn.functional.pad(wr, (0, 0, 0, pad))
    xin = _ParkGradFn.apply(x, slot) if (slot is not None and x.requires_grad) else x
    mixed = proj_bcl(wr, xin.reshape(B, C, H * W), True)                        # (B, rows, HW), stored [rows][B][HW]
    return mix_sample(mixed.view(B, wr.shape[0], H, W), y_rows, O)
