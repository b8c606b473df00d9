"""CPU restatement of the reference's input pipeline arithmetic -- TEST INFRASTRUCTURE ONLY
(imported by tests/ alone; the product path is one_to_many_gan_amd/data/datasets.py +
o2m_gather_images).

Follows train.py:120-126 (``ToTensor`` then ``Normalize((0.5,), (0.5,))``) and
datasets.py:44-50 (``RandomHorizontalFlip`` applied to the normalised tensor), with the
published torchvision definitions of those transforms: ToTensor = uint8 HWC -> float32 CHW / 255;
Normalize = (x - mean) / std; hflip = reverse the last axis.

Parity status: **unpinned against the reference's own loader** -- torchvision is not installed in
this image, so ``src/data/datasets.py`` cannot be imported and the reference holds no fixtures for
it.  The three formulas above are the published ones; the restatement is exact in fp32.
"""

import torch


def to_tensor(u8_hwc: torch.Tensor) -> torch.Tensor:
    return u8_hwc.permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def normalize(t: torch.Tensor, mean=0.5, std=0.5) -> torch.Tensor:
    return (t - mean) / std


def batch(pool_u8_nhwc: torch.Tensor, index, flip) -> torch.Tensor:
    """(B, C, H, W) float32 batch the reference's loader would deliver for these picks."""
    out = []
    for i, f in zip(index.tolist(), flip.tolist()):
        t = normalize(to_tensor(pool_u8_nhwc[i]))
        out.append(t.flip(-1) if f else t)
    return torch.stack(out)
