"""Dice + BCE loss of the reference's top-level ``loss.py`` (:5-28)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

FUSED = os.environ.get("MMUNET_FUSED_LOSS", "1") != "0"   # False: the ATen ops below (fused_paths.plain_aten, tests compare)


class _DiceBceFn(torch.autograd.Function):
    """The whole loss in two launches forward, one backward (csrc/dice_bce.hip); as ATen ops 22 + 14."""

    @staticmethod
    def forward(ctx, logits, targets, smooth):
        from . import _lib
        _lib.require_gpu(logits, targets)
        if logits.dtype != torch.float32 or targets.dtype != torch.float32 or logits.shape != targets.shape:
            raise RuntimeError("dice_bce: float32 logits and targets of one shape required")
        x, t = logits.contiguous(), targets.contiguous()
        n = x.numel()
        L = _lib.lib()
        ws = torch.empty(L.mmu_dice_bce_workspace_floats(n), device=x.device, dtype=torch.float32)
        out = torch.empty(3, device=x.device, dtype=torch.float32)       # loss, sum(p t), sum(p + t)
        p = _lib.DiceBceParams()
        p.n, p.smooth = n, float(smooth)
        p.logits, p.targets, p.workspace, p.out = x.data_ptr(), t.data_ptr(), ws.data_ptr(), out.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(L.mmu_dice_bce_fwd(p, _lib.stream_of(x)))
        ctx.save_for_backward(x, t, out)
        ctx.smooth = float(smooth)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        x, t, out = ctx.saved_tensors
        g = g.float().contiguous()
        dx = torch.empty_like(x)
        p = _lib.DiceBceParams()
        p.n, p.smooth = x.numel(), ctx.smooth
        p.logits, p.targets, p.out, p.dloss, p.dlogits = x.data_ptr(), t.data_ptr(), out.data_ptr(), g.data_ptr(), dx.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mmu_dice_bce_bwd(p, _lib.stream_of(x)))
        return dx, None, None


class DICE_BCE_Loss(nn.Module):
    """``1 - (2*sum(p*t) + s) / (sum(p + t) + s) + BCE(p, t)`` with ``p = sigmoid(logits)``; the Dice
    sums run over the WHOLE batch (loss.py:12-14)."""

    def __init__(self, smooth=1):
        super().__init__()
        self.smooth = smooth

    def forward(self, logits, targets):
        if (FUSED and logits.is_cuda and logits.dtype == torch.float32 and targets.dtype == torch.float32
                and logits.shape == targets.shape and not targets.requires_grad and not torch.is_autocast_enabled()):
            return _DiceBceFn.apply(logits, targets, self.smooth)
        p = torch.sigmoid(logits)
        intersection = 2 * (p * targets).sum() + self.smooth
        union = (p + targets).sum() + self.smooth
        dice_loss = 1.0 - intersection / union
        # BCELoss(sigmoid(x), t) as in loss.py:16-17, evaluated in fp32 (autocast forbids F.binary_cross_entropy
        # on 16-bit probabilities; the value is the same mean BCE)
        bce_loss = F.binary_cross_entropy(p.float(), targets.float())
        return dice_loss + bce_loss


def dice_coeff(logits, targets):
    p = torch.sigmoid(logits)
    intersection = 2 * (p * targets).sum()
    union = (p + targets).sum()
    if union == 0:
        return 1
    return (intersection / union).item()
