"""The validation path of the reference's train.py without MONAI.

train.py:83-139,180-195: ``SlidingWindowInferer(roi_size=image_size, overlap=0.5)`` -> ``sigmoid`` -> threshold 0.5 ->
Dice / mean IoU / F1 / precision / recall / MCC / accuracy accumulated over the validation set.  MONAI is not
installed in the build image, so this is a restatement of its documented algorithms (monai.inferers.sliding_window_inference,
monai.metrics.{DiceMetric, MeanIoU, ConfusionMatrixMetric}; **parity unpinned**: checked against brute-force
definitions in tests/test_validate.py, not against MONAI outputs):
  * windows: scan interval = int(roi * (1 - overlap)) per axis (at least 1), ceil((size - roi) / interval) + 1 windows,
    the last ones clamped to the border (``start = min(i * interval, size - roi)``); an image smaller than the roi is
    zero-padded symmetrically first and the result cropped back; overlapping predictions are averaged with constant
    weights (MONAI's default ``mode="constant"``);
  * confusion-matrix metrics are computed from the counts summed over every validated sample (MONAI's default
    ``compute_sample=False``: counts are reduced first, the metric is taken once);
  * Dice / IoU are per-sample values averaged over the samples where they are defined (MONAI ``ignore_empty=True``:
    a sample with an empty label is skipped).
With DRIVE resized to 608 x 608 and roi 608 (src/VesselLoader.py:278-342, config image_size) every image is exactly
one window, i.e. the inferer is the model's forward.
"""
import math

import torch
import torch.nn.functional as F


def _starts(size, roi, overlap):
    interval = max(int(roi * (1 - overlap)), 1)
    n = max(int(math.ceil((size - roi) / interval)) + 1, 1)
    return sorted({min(i * interval, size - roi) for i in range(n)})


@torch.no_grad()
def sliding_window_inference(images, roi_size, predictor, overlap=0.5, sw_batch_size=1):
    """``predictor(window) -> (B, C, rh, rw)`` over every roi-sized window of ``images`` (B, Cin, H, W), averaged."""
    rh, rw = (roi_size, roi_size) if isinstance(roi_size, int) else tuple(roi_size)
    B, _, H, W = images.shape
    ph, pw = max(rh - H, 0), max(rw - W, 0)
    if ph or pw:
        images = F.pad(images, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    Hp, Wp = images.shape[2:]
    slices = [(y, x) for y in _starts(Hp, rh, overlap) for x in _starts(Wp, rw, overlap)]
    out = cnt = None
    for i in range(0, len(slices), sw_batch_size):
        group = slices[i:i + sw_batch_size]
        win = torch.cat([images[:, :, y:y + rh, x:x + rw] for y, x in group], dim=0)
        pred = predictor(win)
        if out is None:
            out = torch.zeros((B, pred.shape[1], Hp, Wp), device=pred.device, dtype=pred.dtype)
            cnt = torch.zeros((1, 1, Hp, Wp), device=pred.device, dtype=pred.dtype)
        for j, (y, x) in enumerate(group):
            out[:, :, y:y + rh, x:x + rw] += pred[j * B:(j + 1) * B]
            cnt[:, :, y:y + rh, x:x + rw] += 1
    out = out / cnt
    if ph or pw:
        out = out[:, :, ph // 2:ph // 2 + H, pw // 2:pw // 2 + W]
    return out


def post_trans(logits, threshold=0.5):
    """``Activations(sigmoid=True)`` + ``AsDiscrete(threshold=0.5)`` (train.py:192-194)."""
    return (torch.sigmoid(logits) >= threshold).to(logits.dtype)


class SegmentationMetrics:
    """Accumulates what train.py's seven MONAI metrics report (binary masks, channel dimension kept)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.tp = self.fp = self.tn = self.fn = 0.0
        self.dice_sum = self.dice_n = self.iou_sum = self.iou_n = 0.0

    @torch.no_grad()
    def __call__(self, y_pred, y):
        p, t = y_pred.flatten(1).double(), y.flatten(1).double()
        tp, fp = (p * t).sum(1), (p * (1 - t)).sum(1)
        fn, tn = ((1 - p) * t).sum(1), ((1 - p) * (1 - t)).sum(1)
        self.tp += float(tp.sum()); self.fp += float(fp.sum()); self.fn += float(fn.sum()); self.tn += float(tn.sum())
        has_label = t.sum(1) > 0                      # ignore_empty: undefined for an empty label
        dice = 2 * tp / (2 * tp + fp + fn).clamp_min(1e-300)
        iou = tp / (tp + fp + fn).clamp_min(1e-300)
        self.dice_sum += float(dice[has_label].sum()); self.dice_n += float(has_label.sum())
        self.iou_sum += float(iou[has_label].sum()); self.iou_n += float(has_label.sum())

    def aggregate(self):
        tp, fp, tn, fn = self.tp, self.fp, self.tn, self.fn
        div = lambda a, b: a / b if b else float("nan")  # noqa: E731
        mcc_den = math.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn))
        return {
            "dice_metric": div(self.dice_sum, self.dice_n),
            "miou_metric": div(self.iou_sum, self.iou_n),
            "f1": div(2 * tp, 2 * tp + fp + fn),
            "precision": div(tp, tp + fp),
            "recall": div(tp, tp + fn),
            "MCC": div(tp * tn - fp * fn, mcc_den),
            "ACC": div(tp + tn, tp + fp + tn + fn),
        }


@torch.no_grad()
def validate(model, loader, roi_size, loss_fn=None, overlap=0.5):
    """One validation epoch (train.py:83-139): returns (metrics dict, mean loss or None).  ``loader`` yields
    ``(images, labels)`` already on the model's device."""
    model.eval()
    metrics, losses = SegmentationMetrics(), []
    for images, labels in loader:
        logits = sliding_window_inference(images, roi_size, model, overlap)
        if loss_fn is not None:
            losses.append(float(loss_fn(logits, labels)))
        metrics(post_trans(logits), labels)
    return metrics.aggregate(), (sum(losses) / len(losses) if losses else None)
