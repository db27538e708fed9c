"""The DRIVE / STARE pipeline (mm-unet_amd/vessel_loader.py) against direct PIL / torch restatements of every step of
the reference's ``_VesselDatasetInternal._transform`` (src/VesselLoader.py:278-342).  torchvision is not in the image:
parity unpinned, each torchvision call is checked through its documented PIL-path equivalent.  CPU only."""
import os

import numpy as np
import pytest
import torch
from PIL import Image


def _make_tree(root, n=3, size=(584, 565), val_suffix="_manual1"):
    rng = np.random.default_rng(0)
    for phase, suffix in (("train", ""), ("val", val_suffix)):
        os.makedirs(os.path.join(root, phase, "input"))
        os.makedirs(os.path.join(root, phase, "label"))
        for i in range(n):
            img = rng.integers(0, 256, size=(size[0], size[1], 3), dtype=np.uint8)
            lab = (rng.random(size) > 0.85).astype(np.uint8) * 255
            lab[0, :7] = 100          # grey values below the 0.5 threshold must become background
            Image.fromarray(img).save(os.path.join(root, phase, "input", f"{21 + i:02d}_x.png"))
            Image.fromarray(lab).save(os.path.join(root, phase, "label", f"{21 + i:02d}_x{suffix}.png"))
    # an image without a label is skipped, not an error (VesselLoader.py:222-226)
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(os.path.join(root, "val", "input", "99_orphan.png"))


def test_file_discovery_and_val_transform(tmp_path):
    from mm_unet_amd import vessel_loader as vl
    root = str(tmp_path)
    _make_tree(root)
    val = vl.generate_dataset_list(os.path.join(root, "val"), label_filename_pattern=vl.DRIVE["val_label_pattern"])
    assert [os.path.basename(s["image"]) for s in val] == ["21_x.png", "22_x.png", "23_x.png"]
    assert all(s["label"].endswith("_manual1.png") for s in val)
    assert vl.generate_dataset_list(os.path.join(root, "nope")) == []
    ds = vl.VesselDataset(val, "validation", image_size=608)
    x, y, px, py = ds[1]
    assert x.shape == (3, 608, 608) and y.shape == (1, 608, 608) and px == val[1]["image"] and py == val[1]["label"]
    # restatement: 584x565 < 608 -> centre padding (odd pixel right / bottom), PIL bilinear resize (identity here),
    # /255, ImageNet normalisation; label > 0.5 then nearest
    img = np.asarray(Image.open(px).convert("RGB"))
    pt, pl = (608 - 584) // 2, (608 - 565) // 2
    pad = np.zeros((608, 608, 3), np.uint8)
    pad[pt:pt + 584, pl:pl + 565] = img
    ref = torch.from_numpy(pad.transpose(2, 0, 1).copy()).float() / 255
    ref = (ref - torch.tensor(vl.IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(vl.IMAGENET_STD).view(3, 1, 1)
    assert torch.allclose(x, ref, atol=1e-6)
    lab = np.asarray(Image.open(py).convert("L"))
    lpad = np.zeros((608, 608), np.uint8)
    lpad[pt:pt + 584, pl:pl + 565] = lab
    assert torch.equal(y[0], torch.from_numpy((lpad.astype(np.float32) / 255 > 0.5).astype(np.float32)))
    assert set(y.unique().tolist()) <= {0.0, 1.0} and float(y[0, pt, pl]) == 0.0   # the grey 100/255 pixels
    with pytest.raises(IndexError):
        ds[3]


def test_train_transform_resize_and_flips(tmp_path):
    from mm_unet_amd import vessel_loader as vl
    root = str(tmp_path)
    _make_tree(root, size=(700, 640))
    tr = vl.generate_dataset_list(os.path.join(root, "train"), label_filename_pattern=vl.DRIVE["train_label_pattern"])
    ds = vl.VesselDataset(tr, "train", image_size=[608, 512])
    img, lab = Image.open(tr[0]["image"]).convert("RGB"), Image.open(tr[0]["label"]).convert("L")
    seen = set()
    for seed in range(12):
        torch.manual_seed(seed)
        x, y, _, _ = ds[0]
        torch.manual_seed(seed)                      # the same two draws, in the same order (:290-296)
        hf, vf = torch.rand(1).item() > 0.5, torch.rand(1).item() > 0.5
        seen.add((hf, vf))
        im, lb = img, lab
        if hf:
            im, lb = im.transpose(Image.FLIP_LEFT_RIGHT), lb.transpose(Image.FLIP_LEFT_RIGHT)
        if vf:
            im, lb = im.transpose(Image.FLIP_TOP_BOTTOM), lb.transpose(Image.FLIP_TOP_BOTTOM)
        ref = torch.from_numpy(np.asarray(im.resize((512, 608), Image.BILINEAR)).transpose(2, 0, 1).copy()).float() / 255
        ref = (ref - torch.tensor(vl.IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(vl.IMAGENET_STD).view(3, 1, 1)
        assert x.shape == (3, 608, 512) and torch.allclose(x, ref, atol=1e-6)
        lt = (torch.from_numpy(np.asarray(lb).copy()).float() / 255 > 0.5).float()[None, None]
        assert torch.equal(y, torch.nn.functional.interpolate(lt, size=(608, 512), mode="nearest")[0])
    assert len(seen) == 4                            # all four flip combinations occurred


def test_loaders(tmp_path):
    from mm_unet_amd import vessel_loader as vl
    root = str(tmp_path)
    _make_tree(root, n=5, size=(64, 48))
    train, val = vl.get_dataloader(root, batch_size=2, image_size=64, pin_memory=False)
    assert len(train.dataset) == 5 and len(train) == 2 and len(val) == 3      # drop_last in train mode only
    xb, yb, px, py = next(iter(train))
    assert xb.shape == (2, 3, 64, 64) and yb.shape == (2, 1, 64, 64) and len(px) == 2
    order1 = [p for _, _, ps, _ in vl.get_dataloader(root, batch_size=2, image_size=64, pin_memory=False)[0] for p in ps]
    order2 = [p for _, _, ps, _ in vl.get_dataloader(root, batch_size=2, image_size=64, pin_memory=False)[0] for p in ps]
    assert order1 == order2                          # shuffling is driven by the seeded generator (3407)
    assert [os.path.basename(p) for _, _, ps, _ in val for p in ps] == [f"{21 + i:02d}_x.png" for i in range(5)]
    assert vl.get_dataloader(os.path.join(root, "missing"))[0] is None
