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
