"""MI355X-native one-to-many GAN training step.

Drop-in for the hot path of struan-robertson/one-to-many-gan: ``model.builder`` /
``model.layers`` / ``model.blocks`` / ``model.loss`` / ``core.training`` mirror the
reference's ``src.model.*`` and ``src.core.training`` names and signatures; the compute
runs in hand-written gfx950 HIP kernels behind the C ABI of ``include/o2m_hip.h``
(``libo2m_hip.so``, bound with ctypes in ``_hip.py``).  There is no CPU / eager fallback.
"""

from __future__ import annotations

import torch

from . import ops  # noqa: F401
from .ops import compute_dtype, set_deterministic, set_precision  # noqa: F401
from .optim import FusedAdam, make_adam  # noqa: F401

__all__ = ["set_precision", "set_deterministic", "compute_dtype", "make_adam", "FusedAdam", "IdentityADA", "AdaptiveDiscriminatorAugmentation",
           "REFERENCE_ADA_SWITCHES"]

# the switches the reference constructs its augmentation with (train.py:175-188)
REFERENCE_ADA_SWITCHES = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1,
                              contrast=1, lumaflip=1, hue=1, saturation=1)


def __getattr__(name):  # lazy: ada.py pulls in numpy-side operator builders
    if name == "AdaptiveDiscriminatorAugmentation":
        from .ada import AdaptiveDiscriminatorAugmentation

        return AdaptiveDiscriminatorAugmentation
    raise AttributeError(name)


class IdentityADA(torch.nn.Module):
    """The augmentation slot held at p = 0 (reference train.py:175-188, training.py:100,104,200),
    where the pipeline is the identity: what the parity cases and the benchmark use on BOTH sides
    (ADAp starts at p = 0, loss.py:28).  Training runs use ``AdaptiveDiscriminatorAugmentation``
    (ada.py), which implements the transforms for p > 0."""

    def __init__(self, **_kwargs):
        super().__init__()
        self.p = 0.0

    def set_p(self, p: float):
        self.p = float(p)
        if self.p != 0.0:
            raise NotImplementedError("IdentityADA is the p = 0 stand-in: use AdaptiveDiscriminatorAugmentation")

    def forward(self, x):
        return x
