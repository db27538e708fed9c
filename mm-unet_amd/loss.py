"""Dice + BCE loss of the reference's top-level ``loss.py`` (:5-28)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class DICE_BCE_Loss(nn.Module):
    """``1 - (2*sum(p*t) + s) / (sum(p + t) + s) + BCE(p, t)`` with ``p = sigmoid(logits)``; the Dice
    sums run over the WHOLE batch (loss.py:12-14)."""

    def __init__(self, smooth=1):
        super().__init__()
        self.smooth = smooth

    def forward(self, logits, targets):
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
