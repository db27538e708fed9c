"""MONAI-free validation path (mm_unet_amd/validate.py): sliding-window inference against a brute-force average over
the same windows, and the metrics against hand-computed values.  CPU only (any predictor works)."""
import math

import torch


def test_sliding_window_is_the_average_of_its_windows():
    from mm_unet_amd.validate import _starts, sliding_window_inference
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(2, 1, 3, padding=1)
    calls = []

    def predictor(w):
        calls.append(tuple(w.shape))
        return conv(w) + w.mean(dim=(1, 2, 3), keepdim=True)      # depends on the window -> overlaps really differ

    x = torch.randn(2, 2, 21, 30)
    out = sliding_window_inference(x, (8, 12), predictor, overlap=0.5, sw_batch_size=3)
    ys, xs = _starts(21, 8, 0.5), _starts(30, 12, 0.5)
    assert ys == [0, 4, 8, 12, 13] and xs == [0, 6, 12, 18]       # interval 4 / 6, last window clamped to the border
    ref, cnt = torch.zeros(2, 1, 21, 30), torch.zeros(1, 1, 21, 30)
    with torch.no_grad():
        for y in ys:
            for xx in xs:
                ref[:, :, y:y + 8, xx:xx + 12] += predictor(x[:, :, y:y + 8, xx:xx + 12])
                cnt[:, :, y:y + 8, xx:xx + 12] += 1
    assert torch.allclose(out, ref / cnt, atol=1e-6)
    # roi == image: one window = the plain forward (the DRIVE configuration); smaller image: padded and cropped back
    with torch.no_grad():
        assert torch.allclose(sliding_window_inference(x, (21, 30), conv), conv(x), atol=1e-6)
        small = sliding_window_inference(x[:, :, :5, :7], (8, 12), conv)
        assert small.shape == (2, 1, 5, 7)
        padded = torch.nn.functional.pad(x[:, :, :5, :7], (2, 3, 1, 2))
        assert torch.allclose(small, conv(padded)[:, :, 1:6, 2:9], atol=1e-6)


def test_metrics_match_hand_computed_values():
    from mm_unet_amd.validate import SegmentationMetrics, post_trans
    assert post_trans(torch.tensor([[-1.0, 0.0, 2.0]])).tolist() == [[0.0, 1.0, 1.0]]     # sigmoid(0) = 0.5 >= 0.5
    m = SegmentationMetrics()
    # sample 1: tp 2, fp 1, fn 1, tn 4 ; sample 2: empty label (skipped by Dice / IoU), fp 1, tn 7
    p1 = torch.tensor([1, 1, 1, 0, 0, 0, 0, 0.]).view(1, 1, 2, 4)
    t1 = torch.tensor([1, 1, 0, 1, 0, 0, 0, 0.]).view(1, 1, 2, 4)
    p2 = torch.tensor([1, 0, 0, 0, 0, 0, 0, 0.]).view(1, 1, 2, 4)
    t2 = torch.zeros(1, 1, 2, 4)
    m(torch.cat([p1, p2]), torch.cat([t1, t2]))
    r = m.aggregate()
    tp, fp, fn, tn = 2, 2, 1, 11
    assert abs(r["dice_metric"] - 2 * 2 / (2 * 2 + 1 + 1)) < 1e-12 and abs(r["miou_metric"] - 2 / 4) < 1e-12
    assert abs(r["f1"] - 2 * tp / (2 * tp + fp + fn)) < 1e-12
    assert abs(r["precision"] - tp / (tp + fp)) < 1e-12 and abs(r["recall"] - tp / (tp + fn)) < 1e-12
    assert abs(r["ACC"] - (tp + tn) / 16) < 1e-12
    assert abs(r["MCC"] - (tp * tn - fp * fn) / math.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn))) < 1e-12


def test_validate_runs_an_epoch():
    from mm_unet_amd.loss import DICE_BCE_Loss
    from mm_unet_amd.validate import validate
    import mm_unet_amd.unet as pu
    torch.manual_seed(1)
    model = pu.Unet(3, 1)
    data = [(torch.randn(1, 3, 32, 32), (torch.rand(1, 1, 32, 32) > 0.8).float()) for _ in range(2)]
    metrics, loss = validate(model, data, 32, DICE_BCE_Loss())
    assert set(metrics) == {"dice_metric", "miou_metric", "f1", "precision", "recall", "MCC", "ACC"}
    assert loss is not None and math.isfinite(loss) and 0.0 <= metrics["ACC"] <= 1.0


def test_monai_names_of_train_py_run_its_validation_loop():
    """The sys.modules stand-ins for the MONAI names train.py:7,13,180-195,231 uses (mm_unet_amd/monai_shim.py, registered
    by dropin.install() only because MONAI is absent here): the statements of train.py's setup and of val_one_epoch
    (train.py:88-117), written against ``monai.*`` as there, give the values of validate.SegmentationMetrics / brute force."""
    import importlib.util
    import sys
    if "monai" not in sys.modules and importlib.util.find_spec("monai") is not None:
        import pytest
        pytest.skip("a real MONAI is installed: the stand-ins are not registered")
    import mm_unet_amd.dropin
    mm_unet_amd.dropin.install()
    import monai
    from monai.utils import ensure_tuple_rep
    from mm_unet_amd.validate import SegmentationMetrics
    assert ensure_tuple_rep(32, 2) == (32, 32) and ensure_tuple_rep((8, 12), 2) == (8, 12)
    include_background = True
    inference = monai.inferers.SlidingWindowInferer(roi_size=ensure_tuple_rep(16, 2), overlap=0.5, sw_device="cpu",
                                                    device="cpu")
    assert isinstance(inference, monai.inferers.Inferer)
    metrics = {
        'dice_metric': monai.metrics.DiceMetric(include_background=include_background,
                                                reduction=monai.utils.MetricReduction.MEAN_BATCH, get_not_nans=True),
        'miou_metric': monai.metrics.MeanIoU(include_background=include_background, reduction="mean_channel"),
        'f1': monai.metrics.ConfusionMatrixMetric(include_background=include_background, metric_name='f1 score'),
        'precision': monai.metrics.ConfusionMatrixMetric(include_background=include_background, metric_name="precision"),
        'recall': monai.metrics.ConfusionMatrixMetric(include_background=include_background, metric_name="recall"),
        'MCC': monai.metrics.ConfusionMatrixMetric(include_background=include_background,
                                                   metric_name="matthews correlation coefficient"),
        'ACC': monai.metrics.ConfusionMatrixMetric(include_background=include_background, metric_name="accuracy"),
    }
    assert all(isinstance(m, monai.metrics.CumulativeIterationMetric) for m in metrics.values())
    post_trans = monai.transforms.Compose([monai.transforms.Activations(sigmoid=True),
                                           monai.transforms.AsDiscrete(threshold=0.5)])
    torch.manual_seed(3)
    model = torch.nn.Conv2d(3, 1, 3, padding=1)
    loader = [(torch.randn(2, 3, 24, 24), (torch.rand(2, 1, 24, 24) > 0.6).float()) for _ in range(3)]
    loader[1][1][0].zero_()                                   # one sample with an empty label
    mine = SegmentationMetrics()
    first_iou = None
    with torch.no_grad():
        for image_batch in loader:                            # train.py:89-100
            logits = inference(image_batch[0], model)
            val_outputs = post_trans(logits)
            for metric_name in metrics:
                metrics[metric_name](y_pred=val_outputs, y=image_batch[1])
            mine(val_outputs, image_batch[1])
            if first_iou is None:
                p, t = val_outputs[0].flatten(), image_batch[1][0].flatten()
                first_iou = float((p * t).sum() / (p.sum() + t.sum() - (p * t).sum()))
    want = mine.aggregate()
    got = {}
    for metric_name in metrics:                               # train.py:111-117
        batch_acc = metrics[metric_name].aggregate()[0]
        metrics[metric_name].reset()
        got[metric_name] = float(batch_acc.mean())
    for k in ("dice_metric", "f1", "precision", "recall", "MCC", "ACC"):
        assert abs(got[k] - want[k]) < 1e-6, (k, got[k], want[k])
    # MeanIoU(reduction="mean_channel").aggregate() is one value per sample, so train.py's ``[0]`` reads the FIRST sample
    assert abs(got["miou_metric"] - first_iou) < 1e-6
    for m in metrics.values():
        try:
            m.aggregate()
            raise AssertionError("reset() must forget the accumulated batches")
        except ValueError:
            pass
    # DiceFocalLoss as train.py:231 builds it, against the formulas written out
    loss_fn = monai.losses.DiceFocalLoss(smooth_nr=0, smooth_dr=1e-5, to_onehot_y=False, sigmoid=True)
    x, t = torch.randn(2, 1, 6, 5, dtype=torch.float64), (torch.rand(2, 1, 6, 5) > 0.5).double()
    p = torch.sigmoid(x)
    dice = (1 - 2 * (p * t).sum((2, 3)) / (t.sum((2, 3)) + p.sum((2, 3)) + 1e-5)).mean()
    pt = p * t + (1 - p) * (1 - t)
    focal = (-(1 - pt) ** 2 * torch.log(pt)).mean()
    assert abs(float(loss_fn(x, t)) - float(dice + focal)) < 1e-12
    x.requires_grad_(True)
    loss_fn(x, t).backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0
