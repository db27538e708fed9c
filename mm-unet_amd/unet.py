"""Plain U-Net of the reference's top-level ``model.py`` (:5-85): ``Unet(in_channels, classes)`` with
``InConv`` / ``Down`` / ``Up`` / ``OutConv``.  Same sub-module names and creation order, so
state_dicts and seeds are interchangeable.  ATen ops (MIOpen convolutions on the GPU); on the GPU every
BatchNorm2d -> ReLU pair runs as one fused normalisation (norm_fused.bn_act: two passes over the activation
each way instead of five).  This is also the model BASELINE config 1 runs on CPU as a plumbing check (there
the plain modules run)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import conv3x3_mfma, norm_fused


def _conv_bn_relu(seq, x):
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.BatchNorm2d) and norm_fused.bn_act_supported(x, m):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = norm_fused.bn_act(x, m, "relu" if relu else None)
            i += 2 if relu else 1
            continue
        if x.is_cuda and conv3x3_mfma.module_supported(m, x):   # dense 3x3: implicit GEMM on the bf16 matrix cores
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
        else:
            x = m(x)
        i += 1
    return x


class InConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, padding=1), nn.BatchNorm2d(out_channels),
                                  nn.ReLU(inplace=True),
                                  nn.Conv2d(out_channels, out_channels, 3, padding=1), nn.BatchNorm2d(out_channels),
                                  nn.ReLU(inplace=True))

    def forward(self, x):
        return _conv_bn_relu(self.conv, x)


class Down(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.down = nn.Sequential(nn.MaxPool2d(2), InConv(in_channels, out_channels))

    def forward(self, x):
        return self.down(x)


class Up(nn.Module):
    def __init__(self, in_channels, out_channels, bilinear=False):
        super().__init__()
        if bilinear:
            self.up = nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
                                    nn.Conv2d(in_channels, in_channels // 2, 1))
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, 2, stride=2)
        self.conv = InConv(in_channels, out_channels)

    def forward(self, x1, x2):
        x1 = self.up(x1)
        dy = x2.size(2) - x1.size(2)
        dx = x2.size(3) - x1.size(3)
        x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])  # pad-to-match (model.py:40-46)
        return self.conv(torch.cat([x2, x1], dim=1))


class OutConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 1)

    def forward(self, x):
        return self.conv(x)


class Unet(nn.Module):
    def __init__(self, in_channels, classes):
        super().__init__()
        self.n_channels = in_channels
        self.n_classes = classes
        self.inc = InConv(in_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        self.down4 = Down(512, 1024)
        self.up1 = Up(1024, 512)
        self.up2 = Up(512, 256)
        self.up3 = Up(256, 128)
        self.up4 = Up(128, 64)
        self.outc = OutConv(64, classes)

    def forward(self, x):
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        x = self.up1(x5, x4)
        x = self.up2(x, x3)
        x = self.up3(x, x2)
        x = self.up4(x, x1)
        return self.outc(x)
