"""Datasets and the device-resident input pipeline (reference: src/data/datasets.py,
train.py:118-169).

The reference keeps every image in host RAM as a normalised float tensor, flips it per fetch and
feeds the step through ``DataLoader(shuffle=True, drop_last=True, num_workers=8, pin_memory=True)``
wrapped in ``itertools.cycle``.  Here:

* ``ShoeDataset`` / ``Edges2ShoesDataset`` keep the reference's constructor signatures and
  errors (``FileNotFoundError`` on an empty folder) and still work as plain map-style datasets;
  torchvision is not a dependency, so the three transforms train.py composes (``Resize``,
  ``ToTensor``, ``Normalize``) are provided here with the same call conventions.
* ``DeviceImagePool`` holds the resized images ONCE in HBM as uint8 NHWC (a 10 k-image 256x256x3
  set is 1.97 GB of the 288 GB), and ``DeviceLoader`` yields training batches built by one HIP
  launch (``o2m_gather_images``: gather by shuffled index + horizontal flip + ToTensor +
  Normalize, fp32 arithmetic in the reference's order).  A batch is the logical-NCHW view of an
  internal NHWC buffer, so the step functions consume it without a layout pass, worker
  processes, pinned staging or H2D copies.

Sampling follows the reference's loader: a fresh ``torch.randperm(N, generator=g)`` per epoch
(what ``RandomSampler`` draws for ``shuffle=True``), ``drop_last`` batches, endless when wrapped in
``itertools.cycle`` or iterated through ``DeviceLoader.cycle()``.  Flip decisions are
``torch.rand(B, generator=g) < flip_prob`` per batch; the reference draws them inside worker
processes whose seeds depend on the worker id, so its exact flip sequence is not a property of
the algorithm and is not reproduced.
"""

from __future__ import annotations

from pathlib import Path
from typing import Literal

import numpy as np
import torch

from .. import _hip as H
from .. import ops

_dataset_mode = Literal["train", "test", "val"]


# ------------------------------------------------------------------------------ transforms


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


class Resize:
    """``torchvision.transforms.Resize`` on a PIL image: (h, w) gives that exact size, an int
    resizes the shorter edge; bilinear with PIL's antialiasing, as torchvision does for PIL input."""

    def __init__(self, size):
        self.size = size

    def __call__(self, img):
        from PIL import Image

        if isinstance(self.size, int):
            w, h = img.size
            if w <= h:
                new = (self.size, max(1, int(self.size * h / w)))
            else:
                new = (max(1, int(self.size * w / h)), self.size)
        else:
            h, w = self.size
            new = (int(w), int(h))
        return img if img.size == new else img.resize(new, Image.BILINEAR)


def _to_uint8_hwc(img) -> np.ndarray:
    a = np.asarray(img)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8) * 255
    if a.dtype != np.uint8:
        raise TypeError(f"only 8-bit images are supported, got {a.dtype} (mode {getattr(img, 'mode', '?')})")
    return a[:, :, None] if a.ndim == 2 else a


class ToTensor:
    """uint8 HWC (PIL) -> float32 CHW in [0, 1]: ``x / 255``."""

    def __call__(self, img):
        a = torch.from_numpy(np.array(_to_uint8_hwc(img), copy=True))
        return a.permute(2, 0, 1).contiguous().to(torch.float32).div(255)


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = tuple(mean), tuple(std)

    def __call__(self, t: torch.Tensor):
        c = t.shape[0]
        mean = torch.tensor(self.mean if len(self.mean) == c else self.mean * c, dtype=t.dtype).view(c, 1, 1)
        std = torch.tensor(self.std if len(self.std) == c else self.std * c, dtype=t.dtype).view(c, 1, 1)
        return (t - mean) / std


def _is_reference_pipeline(transform) -> bool:
    """Resize -> ToTensor -> Normalize(0.5, 0.5): the pipeline the device pool reproduces."""
    ts = getattr(transform, "transforms", None)
    if not ts or not isinstance(ts[-1], Normalize) or not isinstance(ts[-2] if len(ts) > 1 else None, ToTensor):
        return False
    n = ts[-1]
    return all(m == 0.5 for m in n.mean) and all(s == 0.5 for s in n.std) and \
        all(isinstance(t, Resize) for t in ts[:-2])


# -------------------------------------------------------------------------------- datasets


def _image_files(path: Path):
    files = list(path.rglob("*.jpg")) + list(path.rglob("*.png"))  # reference order: jpg then png
    if len(files) == 0:
        raise FileNotFoundError
    return files


class ShoeDataset(torch.utils.data.Dataset):
    """Load shoe images into RAM (reference datasets.py:13-50)."""

    def __init__(self, path: Path | str, *, mode: _dataset_mode, transform, flip_prob: float = 0.5):
        from PIL import Image

        path = Path(path).expanduser() / mode
        self.transform = transform
        self.flip_prob = flip_prob
        self.raw = []  # uint8 HWC after the geometric part of the transform (for the device pool)
        self.images = []
        resize = [t for t in getattr(transform, "transforms", []) if isinstance(t, Resize)]
        for image_file in _image_files(path):
            image = Image.open(image_file)
            self.images.append(transform(image))
            if _is_reference_pipeline(transform):
                for t in resize:
                    image = t(image)
                self.raw.append(_to_uint8_hwc(image).copy())

    def __len__(self):
        return len(self.images)

    def __getitem__(self, idx: int):
        image = self.images[idx]
        if torch.rand(1) < self.flip_prob:  # RandomHorizontalFlip.forward
            return image.flip(-1)
        return image


class Edges2ShoesDataset(torch.utils.data.Dataset):
    """Paired edges2shoes images, left / right half (reference datasets.py:53-95)."""

    def __init__(self, path: Path | str, *, mode: _dataset_mode, transform, type_: Literal["edge", "shoe"]):
        from PIL import Image

        path = Path(path).expanduser() / mode
        self.transform = transform
        self.flip_prob = 0.0
        self.raw = []
        self.images = []
        resize = [t for t in getattr(transform, "transforms", []) if isinstance(t, Resize)]
        for image_file in _image_files(path):
            image = Image.open(image_file)
            image = image.crop((0, 0, 256, 256)) if type_ == "edge" else image.crop((256, 0, 512, 256))
            self.images.append(transform(image))
            if _is_reference_pipeline(transform):
                for t in resize:
                    image = t(image)
                self.raw.append(_to_uint8_hwc(image).copy())

    def __len__(self):
        return len(self.images)

    def __getitem__(self, idx: int):
        return self.images[idx]


# ------------------------------------------------------------------------ device-resident


class DeviceImagePool:
    """All images of a dataset in HBM as one uint8 [N][H][W][C] tensor."""

    def __init__(self, images, device):
        if isinstance(images, (ShoeDataset, Edges2ShoesDataset)):
            if not images.raw:
                raise ValueError("the dataset's transform is not Resize -> ToTensor -> Normalize(0.5, 0.5): "
                                 "the device pool reproduces exactly that pipeline")
            images = np.stack(images.raw)
        images = torch.as_tensor(images)
        if images.dim() != 4 or images.dtype != torch.uint8:
            raise ValueError("expected uint8 images [N][H][W][C]")
        if torch.device(device).type != "cuda":
            raise RuntimeError("the device-resident pool lives in GPU memory (no CPU fallback)")
        self.data = images.contiguous().to(device)

    def __len__(self):
        return self.data.shape[0]


class DeviceLoader:
    """``DataLoader(dataset, batch_size, shuffle=True, drop_last=True)`` over a DeviceImagePool."""

    def __init__(self, pool: DeviceImagePool, batch_size: int, *, shuffle: bool = True, drop_last: bool = True,
                 flip_prob: float = 0.5, generator: torch.Generator | None = None):
        if batch_size < 1:
            raise ValueError("batch_size should be a positive integer value")
        self.pool, self.batch_size, self.shuffle, self.drop_last = pool, batch_size, shuffle, drop_last
        self.flip_prob, self.generator = flip_prob, generator

    def __len__(self):
        n = len(self.pool)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def plan(self):
        """Host side of one epoch: [(index int32 [b], flip uint8 [b])...] -- the sampler's
        permutation cut into batches, plus the flip decisions."""
        n = len(self.pool)
        order = torch.randperm(n, generator=self.generator) if self.shuffle else torch.arange(n)
        out = []
        for i in range(0, n, self.batch_size):
            idx = order[i: i + self.batch_size]
            if idx.numel() < self.batch_size and self.drop_last:
                break
            flip = torch.rand(idx.numel(), generator=self.generator) < self.flip_prob
            out.append((idx.to(torch.int32), flip.to(torch.uint8)))
        return out

    def __iter__(self):
        data = self.pool.data
        _, h, w, c = data.shape
        for idx, flip in self.plan():
            out = torch.empty((idx.numel(), h, w, ops.pad8(c)), dtype=ops.compute_dtype(), device=data.device)
            H.gather_images(data, idx.to(data.device, non_blocking=True), flip.to(data.device, non_blocking=True), out)
            yield ops.to_public(out, c)

    def cycle(self):
        """``itertools.cycle(loader)`` without caching the first epoch's batches: reshuffles."""
        while True:
            yield from self
