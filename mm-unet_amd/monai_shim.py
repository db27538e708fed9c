"""The MONAI names the reference's ``train.py`` touches, so that its loops run unchanged where MONAI is not installed
(SURVEY.md section 8 row f3).  ``dropin.install()`` registers them as ``sys.modules['monai' ...]`` ONLY when no real
``monai`` package can be found; with MONAI present nothing here is used.

    train.py:13        from monai.utils import ensure_tuple_rep
    train.py:180-181   monai.inferers.SlidingWindowInferer(roi_size=..., overlap=0.5, sw_device=..., device=...)
    train.py:183-191   monai.metrics.DiceMetric / MeanIoU / ConfusionMatrixMetric (+ monai.utils.MetricReduction)
    train.py:193-195   monai.transforms.Compose([Activations(sigmoid=True), AsDiscrete(threshold=0.5)])
    train.py:231       monai.losses.DiceFocalLoss(smooth_nr=0, smooth_dr=1e-5, to_onehot_y=False, sigmoid=True)
    train.py:31,84-86  type annotations: monai.metrics.CumulativeIterationMetric, monai.inferers.Inferer

These are restatements of MONAI's documented behaviour for exactly the argument combinations above (anything else raises):
**parity unpinned** -- MONAI is absent from the build image, so the classes are checked against brute-force definitions and
against ``validate.SegmentationMetrics`` (tests/test_validate.py), not against MONAI outputs.  The shapes of what
``aggregate()`` returns follow MONAI, because train.py indexes them (``aggregate()[0]``, train.py:71,112):
  * DiceMetric(reduction=MEAN_BATCH, get_not_nans=True) -> ``(per-channel mean over the samples where Dice is defined,
    the count of those samples)``; a sample with an empty label is NaN and skipped (``ignore_empty=True``);
  * MeanIoU(reduction="mean_channel") -> ONE tensor with a value per accumulated sample (so train.py's ``[0]`` is the
    first sample's IoU -- the reference's own quirk, kept);
  * ConfusionMatrixMetric(metric_name=one name) -> a list with one scalar tensor, computed from the (tp, fp, tn, fn)
    counts averaged over samples and channels first (``compute_sample=False``).
"""
import enum
import math

import torch
import torch.nn.functional as F

from . import validate


# ---------------------------------------------------------------------------------------------------------- monai.utils
class MetricReduction(str, enum.Enum):
    NONE = "none"
    MEAN = "mean"
    SUM = "sum"
    MEAN_BATCH = "mean_batch"
    SUM_BATCH = "sum_batch"
    MEAN_CHANNEL = "mean_channel"
    SUM_CHANNEL = "sum_channel"


def ensure_tuple_rep(tup, dim):
    """``ensure_tuple_rep(608, 2) -> (608, 608)``; a sequence of length ``dim`` is returned as a tuple."""
    if isinstance(tup, torch.Tensor):
        tup = tup.detach().cpu().tolist()
    if not isinstance(tup, (list, tuple)):
        return (tup,) * dim
    if len(tup) == dim:
        return tuple(tup)
    raise ValueError(f"Sequence must have length {dim}, got {len(tup)}.")


# ------------------------------------------------------------------------------------------------------- monai.inferers
class Inferer:
    def __call__(self, inputs, network, *args, **kwargs):
        raise NotImplementedError


class SlidingWindowInferer(Inferer):
    """``inferer(inputs, network)``: ``network`` over every roi-sized window (scan interval ``int(roi * (1 - overlap))``,
    last windows clamped to the border, constant blending) -- validate.sliding_window_inference."""

    def __init__(self, roi_size, sw_batch_size=1, overlap=0.25, mode="constant", sw_device=None, device=None, **unsupported):
        if str(getattr(mode, "value", mode)) != "constant" or unsupported:
            raise NotImplementedError(f"SlidingWindowInferer shim: mode='constant' only; unsupported {sorted(unsupported)}")
        self.roi_size, self.sw_batch_size, self.overlap = tuple(roi_size), int(sw_batch_size), float(overlap)
        self.sw_device, self.device = sw_device, device

    def __call__(self, inputs, network, *args, **kwargs):
        if self.sw_device is not None:
            inputs = inputs.to(self.sw_device)
        out = validate.sliding_window_inference(inputs, self.roi_size, lambda w: network(w, *args, **kwargs),
                                                self.overlap, self.sw_batch_size)
        return out.to(self.device) if self.device is not None else out


# ----------------------------------------------------------------------------------------------------- monai.transforms
class Compose:
    def __init__(self, transforms=None):
        self.transforms = list(transforms or [])

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class Activations:
    def __init__(self, sigmoid=False, softmax=False, other=None):
        if softmax or other is not None:
            raise NotImplementedError("Activations shim: sigmoid only (train.py:194)")
        self.sigmoid = bool(sigmoid)

    def __call__(self, img):
        return torch.sigmoid(img) if self.sigmoid else img


class AsDiscrete:
    def __init__(self, argmax=False, to_onehot=None, threshold=None, rounding=None):
        if argmax or to_onehot is not None or rounding is not None:
            raise NotImplementedError("AsDiscrete shim: threshold only (train.py:194)")
        self.threshold = threshold

    def __call__(self, img):
        return img if self.threshold is None else (img >= self.threshold).to(img.dtype)


# -------------------------------------------------------------------------------------------------------- monai.metrics
class CumulativeIterationMetric:
    """``metric(y_pred=..., y=...)`` computes the per-sample values of a batch and appends them; ``aggregate()`` reduces
    what has been accumulated; ``reset()`` forgets it."""

    def __init__(self):
        self._buffer = []

    def reset(self):
        self._buffer = []

    def _compute(self, y_pred, y):
        raise NotImplementedError

    def __call__(self, y_pred, y=None):
        if y is None or y_pred.shape != y.shape or y_pred.dim() < 3:
            raise ValueError("y_pred and y must be batch-first tensors (B, C, spatial...) of the same shape")
        v = self._compute(y_pred.detach(), y.detach())
        self._buffer.append(v)
        return v

    def get_buffer(self):
        if not self._buffer:
            raise ValueError("the metric has no accumulated data: call it on a batch first")
        return torch.cat(self._buffer, dim=0)


def _drop_background(include_background, y_pred, y):
    if include_background:
        return y_pred, y
    if y.shape[1] < 2:
        raise ValueError("include_background=False needs more than one channel")
    return y_pred[:, 1:], y[:, 1:]


def _nanmean(t, dim):
    ok = ~torch.isnan(t)
    n = ok.sum(dim)
    s = torch.where(ok, t, torch.zeros_like(t)).sum(dim)
    return torch.where(n > 0, s / n.clamp_min(1), torch.zeros_like(s)), n


class DiceMetric(CumulativeIterationMetric):
    def __init__(self, include_background=True, reduction=MetricReduction.MEAN, get_not_nans=False, ignore_empty=True):
        super().__init__()
        self.include_background, self.get_not_nans, self.ignore_empty = include_background, get_not_nans, ignore_empty
        self.reduction = MetricReduction(getattr(reduction, "value", reduction))
        if self.reduction not in (MetricReduction.MEAN_BATCH, MetricReduction.MEAN):
            raise NotImplementedError("DiceMetric shim: reduction mean_batch (train.py:184) or mean")

    def _compute(self, y_pred, y):
        p, t = _drop_background(self.include_background, y_pred, y)
        p, t = p.flatten(2).double(), t.flatten(2).double()
        inter, yo, po = (p * t).sum(-1), t.sum(-1), p.sum(-1)
        dice = 2 * inter / (yo + po)                         # 0 / 0 -> nan, replaced below
        if self.ignore_empty:
            return torch.where(yo > 0, dice, torch.full_like(dice, float("nan"))).float()
        return torch.where(yo + po > 0, dice, torch.ones_like(dice)).float()

    def aggregate(self):
        f, n = _nanmean(self.get_buffer(), 0)                # over the accumulated samples: one value per channel
        if self.reduction == MetricReduction.MEAN:
            f, n = _nanmean(torch.where(n > 0, f, torch.full_like(f, float("nan"))), 0)
        return (f, n.float()) if self.get_not_nans else f


class MeanIoU(CumulativeIterationMetric):
    def __init__(self, include_background=True, reduction=MetricReduction.MEAN, get_not_nans=False, ignore_empty=True):
        super().__init__()
        self.include_background, self.get_not_nans, self.ignore_empty = include_background, get_not_nans, ignore_empty
        self.reduction = MetricReduction(getattr(reduction, "value", reduction))
        if self.reduction not in (MetricReduction.MEAN_CHANNEL, MetricReduction.MEAN, MetricReduction.MEAN_BATCH):
            raise NotImplementedError("MeanIoU shim: reduction mean_channel (train.py:185), mean or mean_batch")

    def _compute(self, y_pred, y):
        p, t = _drop_background(self.include_background, y_pred, y)
        p, t = p.flatten(2).double(), t.flatten(2).double()
        inter, yo, po = (p * t).sum(-1), t.sum(-1), p.sum(-1)
        union = yo + po - inter
        iou = inter / union
        if self.ignore_empty:
            return torch.where(yo > 0, iou, torch.full_like(iou, float("nan"))).float()
        return torch.where(union > 0, iou, torch.ones_like(iou)).float()

    def aggregate(self):
        buf = self.get_buffer()                              # (samples, channels)
        if self.reduction == MetricReduction.MEAN_CHANNEL:
            f, n = _nanmean(buf, 1)                          # one value per sample
        elif self.reduction == MetricReduction.MEAN_BATCH:
            f, n = _nanmean(buf, 0)
        else:
            f, n = _nanmean(buf, 0)
            f, n = _nanmean(torch.where(n > 0, f, torch.full_like(f, float("nan"))), 0)
        return (f, n.float()) if self.get_not_nans else f


_CM_NAMES = {
    "f1 score": "f1", "f1": "f1", "precision": "precision", "positive predictive value": "precision", "ppv": "precision",
    "recall": "recall", "sensitivity": "recall", "true positive rate": "recall", "tpr": "recall", "hit rate": "recall",
    "accuracy": "accuracy", "acc": "accuracy", "matthews correlation coefficient": "mcc", "mcc": "mcc",
}


def _cm_metric(kind, tp, fp, tn, fn):
    nan = torch.tensor(float("nan"), dtype=tp.dtype, device=tp.device)
    if kind == "f1":
        num, den = 2 * tp, 2 * tp + fp + fn
    elif kind == "precision":
        num, den = tp, tp + fp
    elif kind == "recall":
        num, den = tp, tp + fn
    elif kind == "accuracy":
        num, den = tp + tn, tp + fp + tn + fn
    else:
        num, den = tp * tn - fp * fn, torch.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn))
    return torch.where(den != 0, num / den, nan)


class ConfusionMatrixMetric(CumulativeIterationMetric):
    def __init__(self, include_background=True, metric_name="hit_rate", compute_sample=False,
                 reduction=MetricReduction.MEAN, get_not_nans=False):
        super().__init__()
        names = [metric_name] if isinstance(metric_name, str) else list(metric_name)
        self.kinds = []
        for nme in names:
            key = nme.replace("_", " ").lower()
            if key not in _CM_NAMES:
                raise NotImplementedError(f"ConfusionMatrixMetric shim: metric {nme!r} (have {sorted(set(_CM_NAMES))})")
            self.kinds.append(_CM_NAMES[key])
        if compute_sample or MetricReduction(getattr(reduction, "value", reduction)) != MetricReduction.MEAN or get_not_nans:
            raise NotImplementedError("ConfusionMatrixMetric shim: compute_sample=False, reduction='mean' (train.py:186-191)")
        self.include_background = include_background

    def _compute(self, y_pred, y):
        p, t = _drop_background(self.include_background, y_pred, y)
        p, t = (p.flatten(2) > 0.5).double(), (t.flatten(2) > 0.5).double()
        tp, fp = (p * t).sum(-1), (p * (1 - t)).sum(-1)
        tn, fn = ((1 - p) * (1 - t)).sum(-1), ((1 - p) * t).sum(-1)
        return torch.stack([tp, fp, tn, fn], dim=-1)          # (B, C, 4)

    def aggregate(self):
        cm = self.get_buffer().mean(0).mean(0)               # counts averaged over samples, then channels
        return [_cm_metric(k, cm[0], cm[1], cm[2], cm[3]).float() for k in self.kinds]


# --------------------------------------------------------------------------------------------------------- monai.losses
class DiceFocalLoss(torch.nn.Module):
    """Dice loss + focal loss on logits, as train.py:231 configures it: ``sigmoid=True``, ``to_onehot_y=False``,
    ``smooth_nr``, ``smooth_dr``; MONAI's defaults for the rest (gamma 2, no alpha, both weights 1, ``include_background``,
    mean reduction, non-squared denominators, per-sample Dice):
        dice  = mean_{b,c} [1 - (2 sum(p t) + smooth_nr) / (sum(t) + sum(p) + smooth_dr)],  p = sigmoid(logits)
        focal = mean [ BCE_with_logits(x, t) * (1 - p_t) ** gamma ],                          p_t = p t + (1 - p)(1 - t)"""

    def __init__(self, include_background=True, to_onehot_y=False, sigmoid=False, softmax=False, squared_pred=False,
                 jaccard=False, reduction="mean", smooth_nr=1e-5, smooth_dr=1e-5, batch=False, gamma=2.0,
                 lambda_dice=1.0, lambda_focal=1.0, **unsupported):
        super().__init__()
        if to_onehot_y or softmax or squared_pred or jaccard or batch or not sigmoid or not include_background \
                or reduction != "mean" or unsupported:
            raise NotImplementedError("DiceFocalLoss shim: the configuration of train.py:231 only")
        self.smooth_nr, self.smooth_dr, self.gamma = float(smooth_nr), float(smooth_dr), float(gamma)
        self.lambda_dice, self.lambda_focal = float(lambda_dice), float(lambda_focal)

    def forward(self, input, target):
        if input.shape != target.shape:
            raise ValueError(f"ground truth has different shape ({tuple(target.shape)}) from input ({tuple(input.shape)})")
        target = target.to(input.dtype)
        p = torch.sigmoid(input)
        dims = tuple(range(2, input.dim()))
        inter = (p * target).sum(dims)
        den = target.sum(dims) + p.sum(dims)
        dice = (1.0 - (2.0 * inter + self.smooth_nr) / (den + self.smooth_dr)).mean()
        ce = F.binary_cross_entropy_with_logits(input, target, reduction="none")
        invprobs = F.logsigmoid(-input * (target * 2 - 1))   # log(1 - p_t)
        focal = (ce * torch.exp(invprobs * self.gamma)).mean()
        return self.lambda_dice * dice + self.lambda_focal * focal


def install(put):
    """Registers the module tree through ``put(name, module)`` (dropin.install's setter)."""
    import types

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        m.__dict__.update(attrs)
        m.__doc__ = "mm_unet_amd.monai_shim stand-in (MONAI is not installed): see that module's docstring"
        return put(name, m)

    utils = mod("monai.utils", ensure_tuple_rep=ensure_tuple_rep, MetricReduction=MetricReduction)
    inferers = mod("monai.inferers", Inferer=Inferer, SlidingWindowInferer=SlidingWindowInferer)
    transforms = mod("monai.transforms", Compose=Compose, Activations=Activations, AsDiscrete=AsDiscrete)
    metrics = mod("monai.metrics", CumulativeIterationMetric=CumulativeIterationMetric, DiceMetric=DiceMetric,
                  MeanIoU=MeanIoU, ConfusionMatrixMetric=ConfusionMatrixMetric)
    losses = mod("monai.losses", DiceFocalLoss=DiceFocalLoss)
    return mod("monai", utils=utils, inferers=inferers, transforms=transforms, metrics=metrics, losses=losses,
               __version__="0+mm_unet_amd.shim")
