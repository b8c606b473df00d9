"""MI355X-native one-to-many GAN training step.

Drop-in for the hot path of struan-robertson/one-to-many-gan: ``model.builder`` /
``model.layers`` / ``model.blocks`` / ``model.loss`` / ``core.training`` mirror the
reference's ``src.model.*`` and ``src.core.training`` names and signatures; the compute
runs in hand-written gfx950 HIP kernels behind the C ABI of ``include/o2m_hip.h``
(``libo2m_hip.so``, bound with ctypes in ``_hip.py``).  There is no CPU / eager fallback.
"""

from __future__ import annotations

import torch

from .ops import compute_dtype, set_precision  # noqa: F401
from .optim import FusedAdam, make_adam  # noqa: F401

__all__ = ["set_precision", "compute_dtype", "make_adam", "FusedAdam", "IdentityADA"]


class IdentityADA(torch.nn.Module):
    """Interface slot of ``ada.AdaptiveDiscriminatorAugmentation`` (reference
    train.py:175-188, training.py:100,104,200).  pytorch-ada is an un-vendored dependency
    whose source is not available offline; the hot path is specified, benchmarked and
    parity-tested at augmentation probability p = 0, where the pipeline is the identity."""

    def __init__(self, **_kwargs):
        super().__init__()
        self.p = 0.0

    def set_p(self, p: float):
        self.p = float(p)
        if self.p != 0.0:
            raise NotImplementedError("ADA transforms (p > 0) are outside the built hot path")

    def forward(self, x):
        return x
