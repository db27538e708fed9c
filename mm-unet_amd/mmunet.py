"""MM-UNet (``MM_Net``) and its blocks with the reference's module surface
(src/UM_Net/MMUNet.py): ``MMConv`` (MorphMamba conv, :10-274), ``CBAM`` (:313-338),
``SideoutBlock`` (:341-352), ``RCG`` (:354-418), ``DecoderBlock`` (:420-431),
``ResidualBlock`` (:433-467), ``MM_Net`` (:474-585), ``HPPF`` (:278-310, defined but unused there).

Sub-module names, creation order (RNG draw order) and parameter shapes are the reference's, so
``state_dict`` keys match and equal seeds give equal weights -- including the parameters the
reference creates but never uses (``MMConv.dsc_conv_y``, the ``_b``/``_s`` Mamba branches of "v1"
blocks).  All Mamba compute goes through the HIP kernels; the remaining layers are ATen-ROCm ops.

Differences from the reference, all deliberate:
  * ``MMConv(device=...)`` defaults to ``None`` (build where you are, move with ``.to()``) instead of
    the hard-coded ``"cuda"`` (:19,41-42); coordinate grids are created on the input's device.
  * ``d_state`` is exposed on ``MMConv`` / ``RCG`` / ``MM_Net`` (reference hard-codes 16, :29,355) for
    BASELINE config 5 (d_state=64).
  * the constructor does not print.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .mamba_simple import Mamba, neg_exp, precomputed_A
from . import conv3x3_mfma, conv3x3_small, conv_s2, mamba_small_fused, mfma_gemm, morph_coords, morph_mix, norm_fused
from . import morph_sample as morph_sample_mod
from .morph_sample import morph_sample
from .resize import bilinear_resize
from . import maxpool as maxpool_op, pointwise, stem7
from .tall_gemm import conv1x1_stride2, conv1x1_stride2_supported, dsc_gemm
from .selective_scan_interface import mamba_inner_fn_no_out_proj


def _small_conv3x3(c, input, slot=None):
    """A 3x3 / stride 1 / padding 1 ``nn.Conv2d`` with 1, 2, 6 or 8 output channels through the direct kernels of
    csrc/conv3x3_small.hip (a matrix core has nothing to do there); anything else stays the module's own call."""
    if conv3x3_small.supported(input, c.weight) and c.stride == (1, 1) and c.padding == (1, 1) and \
            c.dilation == (1, 1) and c.groups == 1:
        return conv3x3_small.conv3x3_small(input, c.weight, c.bias, slot)
    return c(input)


class MMConv(nn.Module):
    def __init__(self, in_channels: int = 1, out_channels: int = 1, kernel_size: int = 9, extend_scope: float = 1.0,
                 morph: int = 0, if_offset: bool = True, device=None, num_slices=4, d_state=16):
        super().__init__()
        if morph not in (0, 1):
            raise ValueError("morph should be 0 or 1.")
        self.mamba = Mamba(d_model=kernel_size, d_state=d_state, d_conv=4, expand=2, bimamba_type="v1",
                           nslices=num_slices)
        self.kernel_size = kernel_size
        self.extend_scope = extend_scope
        self.morph = morph
        self.if_offset = if_offset
        if device is not None:
            self.to(device)
        self.gn_offset = nn.GroupNorm(kernel_size, 2 * kernel_size)
        self.gn = nn.GroupNorm(out_channels // 4, out_channels)
        self.relu = nn.ReLU(inplace=False)
        self.tanh = nn.Tanh()
        self.offset_conv = nn.Conv2d(in_channels, 2 * kernel_size, 3, padding=1)
        self.dsc_conv_x = nn.Conv2d(in_channels, out_channels, kernel_size=(kernel_size, 1),
                                    stride=(kernel_size, 1), padding=0)
        self.dsc_conv_y = nn.Conv2d(in_channels, out_channels, kernel_size=(1, kernel_size),
                                    stride=(1, kernel_size), padding=0)
        self.altho = nn.Parameter(torch.log(torch.exp(torch.tensor(1.0)) - 1.0))

    # -- token order of the scan: rows are paired, a pair is walked column by column
    #    (2p,w),(2p+1,w),(2p,w+1)...; an odd last row is appended row-major (MMUNet.py:68-121)
    @staticmethod
    def two_row_columnwise_flatten_grad_safe(x):
        B, C, H, W = x.shape
        He = H // 2 * 2
        flat = x[:, :, :He].reshape(B, C, He // 2, 2, W).permute(0, 1, 2, 4, 3).reshape(B, C, He * W)
        if H % 2 == 1:
            flat = torch.cat([flat, x[:, :, He:].reshape(B, C, -1)], dim=2)
        return flat

    @staticmethod
    def inverse_two_row_columnwise_flatten(x_flat, H, W):
        B, C, _ = x_flat.shape
        He = H // 2 * 2
        out = x_flat[:, :, :He * W].reshape(B, C, He // 2, W, 2).permute(0, 1, 2, 4, 3).reshape(B, C, He, W)
        if H % 2 == 1:
            out = torch.cat([out, x_flat[:, :, He * W:].reshape(B, C, 1, W)], dim=2)
        return out

    def get_coordinate_map_2D(self, offset, morph, extend_scope=1.0, device=None, rows_only=False):
        """Returns (y_map, x_map), each (B, H*K, W): sampling rows / columns of the K taps
        (MMUNet.py:122-193).  Only the row (y) coordinate is learned."""
        if morph not in (0, 1):
            raise ValueError("morph should be 0 or 1.")
        B, _, H, W = offset.shape
        K = offset.shape[1] // 2
        center = K // 2
        dev = offset.device
        y_off = offset[:, :K]  # x_offset (second half) is unused by the reference as well (:136)
        rows = torch.arange(0, H, dtype=torch.float32, device=dev).view(1, 1, H, 1)
        cols = torch.arange(0, W, dtype=torch.float32, device=dev).view(1, 1, 1, W)
        x_spread = torch.linspace(-center, center, K, device=dev).view(1, K, 1, 1)
        x_new = (cols + x_spread).expand(B, K, H, W)
        # iterative offset: centre tap stays, taps further out accumulate (:162-170).  The reference
        # writes these sums (built from the NON-detached offsets) into a detached clone, so gradient
        # does flow through every non-centre tap; only the clone's initial values are cut off.
        acc = [None] * K
        acc[center] = torch.zeros_like(y_off[:, center])
        for i in range(1, center + 1):
            acc[center + i] = acc[center + i - 1] + y_off[:, center + i]
            acc[center - i] = acc[center - i + 1] + y_off[:, center - i]
        y_new = rows + torch.stack(acc, dim=1) * extend_scope
        # Mamba over the offsets, tokens in two-row zig-zag order (:176-183)
        seq = self.two_row_columnwise_flatten_grad_safe(y_off).transpose(-1, -2)
        seq, _, _, _ = self.mamba(seq)
        y_keep = self.inverse_two_row_columnwise_flatten(seq.transpose(-1, -2), H, W)
        weight = torch.clamp(F.softplus(self.altho), min=0.01)
        y = weight * y_keep + y_new
        if rows_only:
            return y  # (B, K, H, W): all the fused sampler needs
        # "b k h w -> b (h k) w"
        y_map = y.permute(0, 2, 1, 3).reshape(B, H * K, W)
        x_map = x_new.permute(0, 2, 1, 3).reshape(B, H * K, W)
        return y_map, x_map

    @staticmethod
    def _coordinate_map_scaling(coordinate_map, origin, target=(-1, 1)):
        lo, hi = origin
        a, b = target
        return a + (b - a) / (hi - lo) * (torch.clamp(coordinate_map, lo, hi) - lo)

    def get_interpolated_feature(self, input_feature, y_coordinate_map, x_coordinate_map, interpolate_mode="bilinear"):
        if interpolate_mode not in ("bilinear", "bicubic"):
            raise ValueError("interpolate_mode should be 'bilinear' or 'bicubic'.")
        y_max = input_feature.shape[-2] - 1
        x_max = input_feature.shape[-1] - 1
        ys = self._coordinate_map_scaling(y_coordinate_map, origin=[0, y_max])
        xs = self._coordinate_map_scaling(x_coordinate_map, origin=[0, x_max])
        grid = torch.stack([xs, ys], dim=-1)  # (B, H*K, W, 2), last dim = (x, y)
        return F.grid_sample(input_feature, grid, mode=interpolate_mode, padding_mode="zeros", align_corners=True)

    def _rows_fused(self, offset, combine=True):
        """get_coordinate_map_2D(rows_only=True) with the tensor glue in two HIP kernels
        (morph_coords): zig-zag + in_proj -> [conv1d, x_proj, dt_proj, selective scan: the uni-directional
        Mamba branch, mamba_simple.py:303-318] -> out_proj + inverse zig-zag + coordinate arithmetic."""
        m = self.mamba
        with torch.autocast("cuda", enabled=False):  # a 2K-channel fp32 scan; grid_sample is fp32 anyway
            if mamba_small_fused.supported(offset, self.kernel_size, m):
                # small maps (16 x 16, 32 x 32): the whole chain below as ONE kernel each way
                return mamba_small_fused.mamba_rows(offset, m, self.altho, self.extend_scope, A=neg_exp(m.A_log),
                                                    combine=combine)
            oslot = morph_coords.OffsetGradSlot() if conv3x3_small.HANDOVER else None   # the two consumers of the offsets hand their d(offset) over
            xz = morph_coords.zigzag_inproj(offset, m.in_proj.weight, oslot)
            out_z = mamba_inner_fn_no_out_proj(xz, m.conv1d.weight, m.conv1d.bias, m.x_proj.weight,
                                               m.dt_proj.weight, neg_exp(m.A_log), None, None,
                                               m.D.float(), delta_bias=m.dt_proj.bias.float(), delta_softplus=True)
            return morph_coords.coords_outproj(offset, out_z, m.out_proj.weight, self.altho, self.extend_scope, oslot)

    def _offset_conv(self, input, slot=None):
        return _small_conv3x3(self.offset_conv, input, slot)   # 6 output channels: direct kernels

    def forward(self, input):
        """GroupNorm(K x 1 DSC conv(deformable samples)) -- MMUNet.py:244-265."""
        pre, bias = self.forward_pre_gn(input)
        if norm_fused.supported(pre, self.gn):
            return norm_fused.gn_bn_act(pre, self.gn, pre_bias=bias, grad_channel_major=self.grad_channel_major(input))
        return self.gn(pre if bias is None else pre + bias.view(1, -1, 1, 1))

    def grad_channel_major(self, input):
        """Layout the block's producer of ``pre`` wants its gradient in: the tokens-last DSC GEMM reads a [C][B][HW]
        gradient in place; the mix-first sampler (and dsc_conv_y) a batch-major one."""
        return self.morph == 0 and not morph_mix.wanted(input, self.dsc_conv_x, self.kernel_size)

    def forward_pre_gn(self, input, slot=None):
        """Everything of forward() before the final GroupNorm (run_fused joins that GroupNorm with the
        BatchNorm2d / ReLU that follow the block in its nn.Sequential).  Returns (conv output WITHOUT its
        bias, bias or None): the fused normalisation folds the bias into its statistics."""
        # the offset convolution and the sampler both read `input`: their input gradients leave as one (GradSlot; a caller
        # whose `input` has a third consumer -- ResidualBlock's shortcut -- passes the slot it parks that gradient in)
        if slot is None:
            slot = conv3x3_small.grad_slot() if input.requires_grad and torch.is_grad_enabled() else None
        raw = self._offset_conv(input, slot)
        if norm_fused.supported(raw, self.gn_offset):
            # GroupNorm -> tanh in 2 passes; float32 also under autocast (coordinates: the reference's group_norm and
            # tanh run in float32 there)
            offset = norm_fused.gn_bn_act(raw, self.gn_offset, None, "tanh", out_dtype=torch.float32)
        else:
            offset = self.tanh(self.gn_offset(raw))
        # Fused HIP sampler (morph_sample): the tap columns are the integers w + k - K//2, so only the
        # row coordinates are passed on.  get_interpolated_feature (grid_sample) stays as the
        # reference-shaped method and is what the fused op is tested against.
        if not morph_sample_mod.ENABLED:
            # the reference's own sequence (MMUNet.py:252-265): coordinate maps, grid_sample, the K x 1 convolution module
            y_map, x_map = self.get_coordinate_map_2D(offset, self.morph, self.extend_scope)
            deformed = self.get_interpolated_feature(input.float() if input.dtype != torch.float32 else input, y_map, x_map)
            return (self.dsc_conv_x if self.morph == 0 else self.dsc_conv_y)(deformed), None
        if self.mamba.in_proj.bias is None and self.mamba.out_proj.bias is None and \
                morph_coords.supported(offset, self.kernel_size):
            y_rows = self._rows_fused(offset, combine=False)   # (the sampler adds the fused kernel's partial maps)
        else:
            y_rows = self.get_coordinate_map_2D(offset, self.morph, self.extend_scope, rows_only=True)
        if self.morph == 0 and morph_mix.wanted(input, self.dsc_conv_x, self.kernel_size):
            # a block that reduces the channel count on a large map: mix the channels first (a 1 x 1 convolution of the
            # input as one GEMM), sample the K * Cout mixed planes afterwards -- the tensor that goes through HBM between
            # the two steps is Cin / Cout times smaller than the sample matrix (morph_mix.py)
            if y_rows.dim() == 5:
                y_rows = y_rows.sum(0)
            return morph_mix.dsc_mix_first(input, y_rows, self.dsc_conv_x, slot), self.dsc_conv_x.bias
        if self.morph == 0:
            # dsc_conv_x (K x 1, stride K x 1) as ONE GEMM: the sampler writes the (Cin*K, B*H*W) matrix
            # [c][k][b][h][w] directly, the conv weight [Cout, Cin, K, 1] viewed as [Cout, Cin*K] multiplies
            # it (split-K weight gradient: the reduction runs over all B*H*W pixels).  MIOpen's implicit GEMM
            # for this conv spent as long in NCHW<->NHWC transposes of the K x inflated tensor as in the math.
            B, _, H, W = input.shape
            conv = self.dsc_conv_x
            samples = morph_sample(input, y_rows, tokens_last=True, slot=slot)
            output = dsc_gemm(conv.weight.view(conv.out_channels, -1), samples, B).view(B, conv.out_channels, H, W)
            return output, conv.bias
        return self.dsc_conv_y(morph_sample(input, y_rows, slot=slot)), None


def run_fused(seq, x, residual=None, in_slot=None, first_slot=None):
    """``seq(x)`` (``relu(seq(x) + residual)`` when a residual is given: the tail of a ResidualBlock) for an
    nn.Sequential, with every ``MMConv -> BatchNorm2d [-> ReLU]`` run as the MMConv up
    to its final GroupNorm followed by ONE fused GroupNorm + BatchNorm + ReLU (norm_fused): 2 passes over the
    activation instead of 8 forward, 2 instead of 13 backward.  Module structure / state_dict are untouched.
    ``in_slot``: a conv3x3_small.SharedGrad of all consumers of ``x``, for a sequence that opens with the stride-2 1 x 1
    shortcut convolution.  ``first_slot``: the conv3x3_small.GradSlot of an opening MMConv when ``residual`` is the same
    tensor as ``x`` (ResidualBlock): the residual's gradient is parked there and added by the sampler's backward kernel
    instead of an autograd add."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, MMConv) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
            pre, bias = m.forward_pre_gn(x, first_slot if i == 0 else None)
            bn = mods[i + 1]
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            if norm_fused.supported(pre, m.gn, bn):
                last = residual is not None and not relu and i + 2 == len(mods)   # ... -> BN, then + residual, ReLU
                x = norm_fused.gn_bn_act(pre, m.gn, bn, "relu" if (relu or last) else None, pre_bias=bias,
                                         grad_channel_major=m.grad_channel_major(x),
                                         residual=conv3x3_small.park_extra(residual, first_slot) if last else None)
                if last:
                    return x
                i += 3 if relu else 2
                continue
            x = m.gn(pre if bias is None else pre + bias.view(1, -1, 1, 1))
            i += 1
            continue
        if isinstance(m, nn.BatchNorm2d) and norm_fused.bn_act_supported(x, m):   # Conv2d -> BatchNorm2d [-> ReLU]
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = norm_fused.bn_act(x, m, "relu" if relu else None)
            i += 2 if relu else 1
            continue
        if conv3x3_mfma.module_supported(m, x):   # dense 3x3 (CBAM): implicit GEMM on the bf16 matrix cores
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if m.bias is not None and isinstance(nxt, nn.BatchNorm2d) and m.bias.dtype == torch.float32:
                # Conv2d(+bias) -> BatchNorm2d [-> ReLU]: the bias is folded into the normalisation
                x = conv3x3_mfma.conv3x3_mfma(x, m.weight, None)
                if norm_fused.bn_act_supported(x, nxt):
                    relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                    x = norm_fused.bn_act(x, nxt, "relu" if relu else None, pre_bias=m.bias)
                    i += 3 if relu else 2
                    continue
                x = x + m.bias.view(1, -1, 1, 1)
            else:
                x = conv3x3_mfma.conv3x3_mfma(x, m.weight, m.bias)
        elif conv1x1_stride2_supported(m, x):     # shortcut of a down-sampling residual block
            x = conv1x1_stride2(x, m.weight, in_slot if i == 0 else None)
        elif conv_s2.module_supported(m, x):      # 3 x 3 / stride 2 opening of a down-sampling residual block
            x = conv_s2.module_call(m, x)
        elif stem7.supported(m, x):               # the 7 x 7 / stride 2 stem on the network's input image
            x = stem7.stem_conv(m, x)
        else:
            x = m(x)
        i += 1
    return x if residual is None else torch.relu(x + residual)


class HPPF(nn.Module):
    """MMUNet.py:278-310 (defined by the reference, not used by MM_Net)."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv2 = nn.Sequential(nn.Conv2d(in_channels, in_channels // 64, 1, 1), nn.ReLU(inplace=True))
        self.conv1 = nn.Sequential(MMConv(in_channels, in_channels // 16, num_slices=64, kernel_size=1),
                                   nn.ReLU(inplace=True))
        self.avg = nn.AdaptiveAvgPool2d(1)
        self.max1 = nn.AdaptiveMaxPool2d(4)
        self.max2 = nn.AdaptiveMaxPool2d(8)
        self.mlp = nn.Sequential(nn.Conv2d(in_channels, in_channels // 8, kernel_size=1), nn.ReLU(inplace=True),
                                 nn.Conv2d(in_channels // 8, in_channels, kernel_size=1), nn.Sigmoid())
        self.feat_conv = nn.Sequential(nn.Conv2d(in_channels, in_channels // 3, 3, 1, 1),
                                       nn.BatchNorm2d(in_channels // 3), nn.ReLU(inplace=True))

    def forward(self, x1, x2, x3):
        x2 = F.interpolate(x2, size=x1.size()[2:], mode="bilinear", align_corners=True)
        x3 = F.interpolate(x3, size=x1.size()[2:], mode="bilinear", align_corners=True)
        feat = torch.cat((x1, x2, x3), 1)
        b, c, h, w = feat.size()
        y1 = self.avg(feat)
        y2 = self.conv1(self.max1(feat)).reshape(b, c, 1, 1)
        y3 = self.conv2(self.max2(feat)).reshape(b, c, 1, 1)
        attention = self.mlp((y1 + y2 + y3) / 3)
        return self.feat_conv(attention * feat)


class CBAM(nn.Module):
    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.max_pool = nn.AdaptiveMaxPool2d(1)
        self.mlp = nn.Sequential(nn.Conv2d(channel, channel // reduction, kernel_size=1, bias=False),
                                 nn.ReLU(inplace=True),
                                 nn.Conv2d(channel // reduction, channel, kernel_size=1, bias=False))
        self.conv = nn.Conv2d(2, 1, kernel_size=7, stride=1, padding=3, bias=False)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        # global max as a flat reduction: ATen's adaptive_max_pool2d kernel takes 6 ms on the
        # [8, 64, 256, 256] stem map (one thread per output element); same value, and like the pooling
        # op the gradient goes to one arg-max element.
        fused = pointwise.stats_supported(x)
        # hand-overs of the backward pass (no activation-sized gradient adds): x feeds the pooled statistics and the
        # first product; y1 the spatial statistics and the second product
        share = fused and pointwise.GATED_MUL and torch.is_grad_enabled() and x.requires_grad
        xs = conv3x3_small.shared_grad() if share else None
        ss = pointwise.ChannelStatsSlot() if (share and conv3x3_small.HANDOVER) else None
        if fused:    # mean and max (+ arg-max) in one pass each way (csrc/cbam_stats.hip)
            x_avg, x_max = pointwise.pixel_mean_max(x, xs)
        else:
            x_avg = self.avg_pool(x)
            x_max = x.flatten(2).max(dim=2)[0].unsqueeze(-1).unsqueeze(-1)
        if fused and pointwise.cbam_gate_supported(self.mlp, x_avg):    # the shared MLP + sigmoid: one launch each way
            c_out = pointwise.cbam_gate(self.mlp, x_avg, x_max)
        else:
            c_out = self.sigmoid(self.mlp(x_avg) + self.mlp(x_max))
        y1 = pointwise.gated_mul(x, c_out, xs, ss)
        if fused:
            s_cat = pointwise.channel_max_mean(y1, ss)
        else:
            s_avg = torch.mean(y1, dim=1, keepdim=True)
            s_max, _ = torch.max(y1, dim=1, keepdim=True)
            s_cat = torch.cat((s_max, s_avg), 1)
        s_out = self.sigmoid(pointwise.conv7_module(self.conv, s_cat))
        return pointwise.gated_mul(y1, s_out)


class SideoutBlock(nn.Module):
    def __init__(self, in_channels, out_channels, num_slices=4, d_state=16):
        super().__init__()
        self.conv1 = nn.Sequential(MMConv(in_channels, in_channels // 4, num_slices=num_slices, kernel_size=3,
                                          d_state=d_state),
                                   nn.BatchNorm2d(in_channels // 4), nn.ReLU(inplace=True))
        self.dropout = nn.Dropout2d(0.1)
        self.conv2 = nn.Conv2d(in_channels // 4, out_channels, kernel_size=1)

    def forward(self, x):
        return pointwise.conv_module(self.conv2, run_fused(self.conv1, x), dropout=self.dropout)


class RCG(nn.Module):
    def __init__(self, d_state=16, d_conv=4, expand=2, head=4, num_slices=4, step=1):
        super().__init__()
        self.conv1 = nn.Sequential(MMConv(128, 64, num_slices=num_slices, kernel_size=3, d_state=d_state),
                                   nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        self.upsample = nn.ConvTranspose2d(64, 64, kernel_size=4, stride=2, padding=1, output_padding=0)
        self.downsample = nn.Conv2d(64, 64, kernel_size=4, stride=2, padding=1)
        self.mamba = Mamba(d_model=64, d_state=d_state, d_conv=d_conv, expand=expand, bimamba_type="v3",
                           nslices=num_slices)
        self.mamba.return_branch_outputs = False   # forward() below keeps only the block output (MMUNet.py:409)
        self.mlp = nn.Sequential(nn.Conv2d(64, 1, kernel_size=1), nn.Sigmoid())

    def forward(self, pre, edge, f, edge_slot=None):
        """``edge_slot``: a conv3x3_small.SharedGrad of all consumers of ``edge`` (MM_Net: three RCG blocks + the line head)."""
        r = pointwise.gated_mul(f, 1 - torch.sigmoid(pre))     # (1 - sigmoid(pre)) * f with a per-pixel gate: one pass each way
        edge1 = bilinear_resize(edge, size=f.size()[2:], slot=edge_slot)
        x2 = run_fused(self.conv1, torch.cat((edge1, r), 1))
        # tri-directional Mamba at 2x resolution (MMUNet.py:398-412)
        x0 = conv_s2.module_call(self.upsample, x2)      # ConvTranspose2d(64, 64, 4, 2, 1) on the matrix cores
        B, C, H, W = x0.shape
        # (channels-first entry: same block as self.mamba(x0.flatten(2).transpose(1, 2)), without the
        # (B, L, C) round trip -- 134 MB transposing copies each way at 256 x 256)
        out, _, _, _ = self.mamba.forward_bcl(x0.reshape(B, C, H * W))
        x0 = conv_s2.module_call(self.downsample, out.reshape(B, C, H, W))
        gate = torch.sigmoid(pointwise.conv_module(self.mlp[0], x2))       # mlp = Conv2d(64, 1, 1) -> Sigmoid
        return pointwise.gated_mul3(x0, x2, gate, f)           # x0 * x2 * gate + f in one pass


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, out_channels, num_slices=4, d_state=16):
        super().__init__()
        self.conv1 = nn.Sequential(MMConv(in_channels, in_channels // 4, kernel_size=3, num_slices=num_slices,
                                          d_state=d_state),
                                   nn.BatchNorm2d(in_channels // 4), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(MMConv(in_channels // 4, out_channels, kernel_size=3, num_slices=num_slices,
                                          d_state=d_state),
                                   nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def forward(self, x):
        return bilinear_resize(run_fused(self.conv2, run_fused(self.conv1, x)), scale_factor=2)


class ResidualBlock(nn.Module):
    def __init__(self, in_channels, out_channels, num_slices, downsample=False, d_state=16):
        super().__init__()
        self.downsample = downsample
        if downsample:
            self.block1 = nn.Sequential(
                nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=2, padding=1, bias=False),
                nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True),
                MMConv(out_channels, out_channels, num_slices=num_slices, kernel_size=3, d_state=d_state),
                nn.BatchNorm2d(out_channels))
            self.block2 = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=2, bias=False),
                                        nn.BatchNorm2d(out_channels))
        else:
            self.block1 = nn.Sequential(
                MMConv(in_channels, out_channels, num_slices=num_slices, kernel_size=3, d_state=d_state),
                nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True),
                MMConv(out_channels, out_channels, num_slices=num_slices, kernel_size=3, d_state=d_state),
                nn.BatchNorm2d(out_channels))
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        # relu(block1(x) + shortcut): the add and the ReLU ride along in block1's last fused normalisation
        if not self.downsample:
            # x feeds block1's first MMConv (offset convolution + sampler: one GradSlot) and the residual add: the
            # residual's gradient rides along in that slot instead of being added by autograd
            slot = conv3x3_small.grad_slot() if (torch.is_grad_enabled() and x.requires_grad) else None
            return run_fused(self.block1, x, residual=x, first_slot=slot)
        # x feeds the 3 x 3 stride-2 convolution and the 1 x 1 stride-2 shortcut: the shortcut's backward adds its even
        # pixels to the gradient the other one left (conv3x3_small.SharedGrad) instead of autograd adding two tensors
        slot = conv3x3_small.shared_grad() if (torch.is_grad_enabled() and x.requires_grad) else None
        shortcut = run_fused(self.block2, x, in_slot=slot)
        return run_fused(self.block1, conv3x3_small.shared_input(x, slot), residual=shortcut)


class MM_Net(nn.Module):
    def __init__(self, num_classes, num_slices_list=[64, 32, 16, 8], out_indices=[0, 1, 2, 3], heads=[1, 2, 4, 4],
                 d_state=16):
        super().__init__()
        s = num_slices_list
        rb = lambda i, o, n, **kw: ResidualBlock(i, o, num_slices=n, d_state=d_state, **kw)  # noqa: E731
        self.encoder1 = nn.Sequential(nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False),
                                      nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1, dilation=1, ceil_mode=False)
        self.encoder2 = nn.Sequential(rb(64, 64, s[0]), rb(64, 64, s[0]), rb(64, 64, s[0]))
        self.encoder3 = nn.Sequential(rb(64, 128, s[1], downsample=True), rb(128, 128, s[1]), rb(128, 128, s[1]),
                                      rb(128, 128, s[1]))
        self.encoder4 = nn.Sequential(rb(128, 256, s[2], downsample=True), rb(256, 256, s[2]), rb(256, 256, s[2]),
                                      rb(256, 256, s[2]), rb(256, 256, s[2]), rb(256, 256, s[2]))
        self.encoder5 = nn.Sequential(rb(256, 512, s[3], downsample=True), rb(512, 512, s[3]), rb(512, 512, s[3]))

        def down(c):
            return nn.Sequential(MMConv(c, 64, num_slices=s[-1], kernel_size=1, d_state=d_state),
                                 nn.BatchNorm2d(64), nn.ReLU(inplace=True))

        self.down3 = down(128)
        self.down4 = down(256)
        self.down5 = down(512)
        self.cbam = nn.Sequential(nn.Conv2d(64, 64, 3, 1, 1), nn.BatchNorm2d(64), nn.ReLU(inplace=True), CBAM(64),
                                  nn.Conv2d(64, 64, 3, 1, 1), nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        self.line_predict = nn.Conv2d(64, 1, 3, 1, 1)
        self.side2 = SideoutBlock(64, 1, num_slices=s[0], d_state=d_state)
        self.side3 = SideoutBlock(64, 1, num_slices=s[1], d_state=d_state)
        self.side4 = SideoutBlock(64, 1, num_slices=s[2], d_state=d_state)
        self.side5 = SideoutBlock(64, 1, num_slices=s[3], d_state=d_state)
        self.rcg2 = RCG(d_state=d_state, num_slices=s[0], head=heads[0])
        self.rcg3 = RCG(d_state=d_state, num_slices=s[1], head=heads[1])
        self.rcg4 = RCG(d_state=d_state, num_slices=s[2], head=heads[2])
        self.decoder5 = DecoderBlock(in_channels=64, out_channels=64, num_slices=s[3], d_state=d_state)
        self.decoder4 = DecoderBlock(in_channels=128, out_channels=64, num_slices=s[2], d_state=d_state)
        self.decoder3 = DecoderBlock(in_channels=128, out_channels=64, num_slices=s[1], d_state=d_state)
        self.decoder2 = DecoderBlock(in_channels=128, out_channels=64, num_slices=s[0], d_state=d_state)

    def forward(self, x):
        if getattr(self, "_a_batch", None) is None:
            self._a_batch = [precomputed_A(self)]   # in a list: not a sub-module, just the parameter list found once
            # the K x 1 DSC weights as the (Cout, Cin*K) matrices dsc_gemm multiplies with
            blocks = [m for m in self.modules() if isinstance(m, MMConv) and m.morph == 0]
            rcgs = [m.mamba for m in self.modules() if isinstance(m, RCG)]   # their projections take the same GEMM
            self._dsc_prep = [mfma_gemm.prepared_weights(
                lambda: [m.dsc_conv_x.weight for m in blocks] + [w for m in rcgs for w in (m.in_proj.weight, m.out_proj.weight)]
                + [getattr(m, n).weight for m in rcgs for n in ("x_proj", "x_proj_b", "x_proj_s") if hasattr(m, n)]
                + [m.block2[0].weight for m in self.modules() if isinstance(m, ResidualBlock) and m.downsample])]
        # A = -exp(A_log) of all 50 Mamba blocks in two launches; the 56 BatchNorm batch counters in one; the bf16 hi/lo
        # images of the 47 DSC weights (both orientations) in one
        with self._a_batch[0], norm_fused.batched_counters(), self._dsc_prep[0]:
            return self._forward(x)

    def _forward(self, x):
        size = x.size()[2:]
        up = lambda t: bilinear_resize(t, size=size)  # noqa: E731
        e1 = run_fused(self.encoder1, x)
        # e1 feeds the max-pool and CBAM, the edge map c1 three RCG blocks and the line head: their input gradients are
        # handed from consumer to consumer (conv3x3_small.SharedGrad) instead of added by autograd
        share = torch.is_grad_enabled() and e1.requires_grad
        s1 = conv3x3_small.shared_grad() if share else None
        e2 = self.encoder2(maxpool_op.pool_module(self.maxpool, e1, s1))
        e3 = self.encoder3(e2)
        e4 = self.encoder4(e3)
        e5 = self.encoder5(e4)
        e3, e4, e5 = run_fused(self.down3, e3), run_fused(self.down4, e4), run_fused(self.down5, e5)
        d5 = self.decoder5(e5)
        out5 = self.side5(d5)
        c1 = run_fused(self.cbam, conv3x3_small.shared_input(e1, s1))   # contour branch on the stride-2 stem features
        sc = conv3x3_small.shared_grad() if share else None
        p_c = _small_conv3x3(self.line_predict, c1, sc)
        r4 = self.rcg4(out5, c1, e4, sc)
        d4 = self.decoder4(torch.cat((d5, r4), dim=1))
        out4 = self.side4(d4)
        r3 = self.rcg3(out4, c1, e3, sc)
        d3 = self.decoder3(torch.cat((d4, r3), dim=1))
        out3 = self.side3(d3)
        r2 = self.rcg2(out3, c1, e2, sc)
        d2 = self.decoder2(torch.cat((d3, r2), dim=1))
        out2 = self.side2(d2)
        return up(out2) + up(out3) + up(out4) + up(out5) + up(p_c)
