"""DRIVE / STARE data pipeline with the semantics of the reference's ``src/VesselLoader.py``,
without torchvision / easydict (neither is in the image): PIL + torch only.

What the reference does per sample (``_VesselDatasetInternal._transform``, VesselLoader.py:278-342) and what is
mirrored here line by line:

  * all images are decoded once at construction (``.convert('RGB')`` / ``.convert('L')``, :262-266);
  * validation / test: an image smaller than ``image_size`` is centre-padded with zeros (:283-287, ``center_padding``
    :103-141: ``pad_left = (W - w) // 2``, the rest on the right; same for top / bottom);
  * train: horizontal flip if ``torch.rand(1) > 0.5``, then vertical flip if ``torch.rand(1) > 0.5`` -- image and label
    together, in that draw order (:290-296);
  * image: ``Resize([H, W], antialias=True)`` on the PIL image (= ``PIL.Image.resize((W, H), BILINEAR)``),
    ``ToTensor`` (CHW float32 / 255), ``Normalize(mean, std)`` (:312-316);
  * label: ``to_tensor`` (/255), ``> 0.5`` -> {0, 1} float, then nearest-neighbour resize of the TENSOR to [H, W]
    (``TF.resize(..., NEAREST)`` = ``F.interpolate(mode='nearest')``, :335-340);
  * ``__getitem__`` returns ``(image, label, image_path, label_path)`` (:344-352);
  * file discovery: every file of ``<root>/<phase>/<image_subdir>``, sorted, paired with
    ``label_filename_pattern.format(base_name=...)`` in ``<label_subdir>``; images without a label are skipped
    (``generate_dataset_list``, :196-230); DRIVE patterns ``{base_name}.png`` (train) / ``{base_name}_manual1.png`` (val),
    STARE ``{base_name}.ah.ppm`` (:410-416);
  * loader: batch from the config, shuffle + drop_last in train mode, generator seeded with ``random_seed`` (3407 by
    default), ``seed_worker`` (:355-383).

The optional augmentations the shipped config leaves off (cut-mix, random-resized-crop, colour jitter, blur,
:298-330) are not rebuilt.  torchvision is absent, so its calls are restated from their documented PIL paths:
**parity unpinned** (tests/test_vessel_loader.py checks every step against direct PIL / torch computations).
"""
import os
import random

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image
from torch.utils.data import DataLoader, Dataset

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # config.yml:27-28,35-36
IMAGENET_STD = (0.229, 0.224, 0.225)
DRIVE = dict(image_size=608, train_label_pattern="{base_name}.png", val_label_pattern="{base_name}_manual1.png")
STARE = dict(image_size=704, train_label_pattern="{base_name}.ah.ppm", val_label_pattern="{base_name}.ah.ppm")


def generate_dataset_list(phase_root, image_subdir="input", label_subdir="label", label_filename_pattern="{base_name}.png"):
    """[{'image': path, 'label': path}, ...] (VesselLoader.py:196-230)."""
    image_dir, label_dir = os.path.join(phase_root, image_subdir), os.path.join(phase_root, label_subdir)
    if not os.path.isdir(image_dir) or not os.path.isdir(label_dir):
        return []
    out = []
    for name in sorted(os.listdir(image_dir)):
        base, _ = os.path.splitext(name)
        label = os.path.join(label_dir, label_filename_pattern.format(base_name=base))
        if os.path.exists(label):
            out.append({"image": os.path.join(image_dir, name), "label": label})
    return out


def _to_tensor(img):
    """torchvision ``to_tensor`` for 8-bit PIL images: HWC uint8 -> CHW float32 / 255."""
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(a.transpose(2, 0, 1).copy()).float().div(255.0)


def center_padding(img, target_size, pad_digit=0):
    """VesselLoader.py:103-141 for PIL images: zero padding to at least ``target_size`` = [H, W], the odd pixel on the
    right / bottom; returned as a PIL image of the original mode."""
    w, h = img.size
    th, tw = target_size
    if h >= th and w >= tw:
        return img
    pl = max(0, (tw - w) // 2)
    pr = max(0, tw - w - pl)
    pt = max(0, (th - h) // 2)
    pb = max(0, th - h - pt)
    a = np.asarray(img, dtype=np.uint8)
    pads = ((pt, pb), (pl, pr)) + (((0, 0),) if a.ndim == 3 else ())
    # (the reference pads the /255 tensor and converts back with to_pil_image: value pad_digit * 255, clipped to a byte)
    return Image.fromarray(np.pad(a, pads, mode="constant", constant_values=int(min(max(pad_digit, 0), 1) * 255)), mode=img.mode)


class VesselDataset(Dataset):
    """``_VesselDatasetInternal`` (VesselLoader.py:233-352)."""

    def __init__(self, samples, mode, image_size=608, image_mean=IMAGENET_MEAN, image_std=IMAGENET_STD, loader=Image.open):
        assert mode in ("train", "validation", "test")
        self.mode = mode
        self.image_size = [image_size, image_size] if isinstance(image_size, int) else list(image_size)
        self.mean = torch.tensor(image_mean, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(image_std, dtype=torch.float32).view(3, 1, 1)
        self.img_paths_x = [s["image"] for s in samples]
        self.img_paths_y = [s["label"] for s in samples]
        self.images_x = [loader(p).convert("RGB") for p in self.img_paths_x]   # mounted on memory, :256-270
        self.images_y = [loader(p).convert("L") for p in self.img_paths_y]

    def __len__(self):
        return len(self.images_x)

    def _transform(self, image, target):
        th, tw = self.image_size
        if self.mode in ("validation", "test"):
            w, h = image.size
            if h < th or w < tw:
                image, target = center_padding(image, [th, tw]), center_padding(target, [th, tw])
        if self.mode == "train":
            if torch.rand(1).item() > 0.5:
                image, target = image.transpose(Image.FLIP_LEFT_RIGHT), target.transpose(Image.FLIP_LEFT_RIGHT)
            if torch.rand(1).item() > 0.5:
                image, target = image.transpose(Image.FLIP_TOP_BOTTOM), target.transpose(Image.FLIP_TOP_BOTTOM)
        x = _to_tensor(image.resize((tw, th), Image.BILINEAR))
        x = (x - self.mean) / self.std
        y = (_to_tensor(target if target.mode == "L" else target.convert("L")) > 0.5).float()
        y = F.interpolate(y[None], size=(th, tw), mode="nearest")[0]
        return x, y

    def __getitem__(self, index):
        if index >= len(self.images_x):
            raise IndexError(f"Index {index} out of bounds for dataset of size {len(self.images_x)}")
        x, y = self._transform(self.images_x[index], self.images_y[index])
        return x, y, self.img_paths_x[index], self.img_paths_y[index]


def seed_worker(worker_id):
    """VesselLoader.py:144-147."""
    s = torch.initial_seed() % 2 ** 32
    np.random.seed(s)
    random.seed(s)


def make_loader(samples, mode, batch_size=4, num_workers=0, pin_memory=True, random_seed=3407, **dataset_kw):
    """``Image2ImageLoader_zero_pad(...).Loader`` (VesselLoader.py:355-383)."""
    g = torch.Generator()
    g.manual_seed(random_seed)
    return DataLoader(VesselDataset(samples, mode, **dataset_kw), batch_size=batch_size, shuffle=(mode == "train"),
                      num_workers=num_workers, worker_init_fn=seed_worker, generator=g, pin_memory=pin_memory,
                      drop_last=(mode == "train"))


def get_dataloader(data_root, batch_size=5, num_workers=0, image_size=608, image_mean=IMAGENET_MEAN, image_std=IMAGENET_STD,
                   train_dir="train", val_dir="val", image_subdir="input", label_subdir="label",
                   train_label_pattern="{base_name}.png", val_label_pattern="{base_name}_manual1.png", random_seed=3407,
                   pin_memory=True):
    """``get_dataloader(config)`` (VesselLoader.py:388-480) with the dataset block of config.yml as keyword arguments
    (defaults = the DRIVE block, config.yml:22-28; pass ``**STARE`` for STARE).  Returns ``(train_loader, val_loader)``,
    ``None`` for a phase whose directory or samples are missing."""
    kw = dict(image_size=image_size, image_mean=image_mean, image_std=image_std)
    loaders = []
    for phase_dir, mode, pattern in ((train_dir, "train", train_label_pattern), (val_dir, "validation", val_label_pattern)):
        samples = generate_dataset_list(os.path.join(data_root, phase_dir), image_subdir, label_subdir, pattern)
        loaders.append(make_loader(samples, mode, batch_size, num_workers, pin_memory, random_seed, **kw) if samples else None)
    return tuple(loaders)
