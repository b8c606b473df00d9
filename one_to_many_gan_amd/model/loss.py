"""Losses and the ADA probability controller with the reference's signatures
(src/model/loss.py).  The image/feature-sized reductions (KL moments, path energy) run as
HIP reduction kernels on the NHWC buffers; B x w_dim arithmetic stays in torch."""

from __future__ import annotations

import torch
from torch.nn import functional as F

from .. import ops


class ADAp:
    """Controller of the augmentation probability (reference loss.py:11-52).

    Every ``window = ada_e // batch_size`` discriminator steps the mean of the collected
    ``sign(D(real))`` scores is compared with the target: below it p drops by
    ``ada_adjustment_size * ada_e``, above it p rises by the same amount, and p never goes
    negative (there is no upper clamp in the reference either).  Reference quirk kept: the score
    that closes a window is counted in that window AND opens the next one (loss.py:34,49).
    Scores stay device tensors; only ``__call__`` (one ``.item()``) synchronises.
    ``p`` / ``curr_batch`` / ``mean_real_scores`` are the state the checkpoint code saves."""

    def __init__(self, ada_e: float, ada_adjustment_size: float, batch_size: int,
                 discriminator_overfitting_target: float):
        self.window, self.step, self.target = ada_e // batch_size, ada_adjustment_size * ada_e, \
            discriminator_overfitting_target
        self.p = torch.zeros(())
        self.curr_batch, self.mean_real_scores = 0, []

    def _adjust(self):
        scores = torch.stack([t.detach().float().cpu() for t in self.mean_real_scores])
        verdict = scores.mean()
        delta = self.step if verdict > self.target else (-self.step if verdict < self.target else 0.0)
        self.p = torch.clamp_min(self.p + delta, 0.0)

    def update_p(self, mean_score: torch.Tensor):
        self.mean_real_scores.append(mean_score)
        if self.curr_batch == self.window:
            self._adjust()
            self.mean_real_scores, self.curr_batch = [mean_score], 0
        self.curr_batch += 1

    def __call__(self) -> float:
        return self.p.item()


def style_cycle_loss_func(original_w: torch.Tensor, reconstructed_w: torch.Tensor, *,
                          normalise=True, cos_l2_ratio: float = 0.2):
    """1 - mean cosine + ratio * MSE on (B, w_dim) vectors (reference loss.py:60-75)."""
    a, b = original_w.float(), reconstructed_w.float()
    if normalise:
        a, b = F.normalize(a, dim=-1), F.normalize(b, dim=-1)
    return 1 - F.cosine_similarity(a, b, dim=-1).mean() + cos_l2_ratio * F.mse_loss(a, b)


def kl_loss_func(combined_latents: torch.Tensor, *, moment_hook=None):
    """mean^2 + (biased var - 1)^2 over the WHOLE latent batch (reference loss.py:82-92),
    from one fused two-moment reduction.  ``moment_hook(s1, s2, n)`` lets data-parallel
    training all-reduce the two sums so the loss is that of the global batch."""
    n = combined_latents.numel()
    s1, s2 = ops.moments(ops.to_internal(combined_latents))
    if moment_hook is not None:
        s1, s2, n = moment_hook(s1, s2, n)
    mean = s1 / n
    var = s2 / n - mean * mean
    return mean**2 + (var - 1) ** 2


def path_loss_func(features1: list[torch.Tensor], features2: list[torch.Tensor],
                   cent_fin_diff_h: torch.Tensor) -> torch.Tensor:
    """Mean over maps of mean(((f1-f2)/h_b)^2) (reference loss.py:98-111); one weighted
    squared-difference reduction per map with w_b = 1/h_b^2."""
    if len(features1) != len(features2):
        raise ValueError("zip() argument lengths differ")
    inv_h2 = (1.0 / (cent_fin_diff_h.float() ** 2)).contiguous()
    total = torch.zeros((), device=features1[0].device)
    for f1, f2 in zip(features1, features2):
        total = total + ops.sq_sum(ops.to_internal(f1), ops.to_internal(f2), inv_h2) / f1.numel()
    return total / len(features1)


def path_loss_halves(features, cent_fin_diff_h: torch.Tensor) -> torch.Tensor:
    """``path_loss_func`` for the fused step: ``features`` = [(internal NHWC buffer [2B, H, W, Cp], logical
    channel count)] of ONE 2B extraction pass whose first half used the style w(theta + h/2) and second half
    w(theta - h/2) (core/training.py).  Same value as path_loss_func(first halves, second halves, h) (reference
    loss.py:98-111); padded channels are zero in both halves and do not count in the mean."""
    inv_h2 = (1.0 / (cent_fin_diff_h.float() ** 2)).contiguous()
    terms, coefs = [], []
    for entry in features:
        if len(entry) == 3:  # (pair term taken while the map passed, channels, shape): Generator._decode(tap=...)
            term, c, (b2, hh, ww, _) = entry
        else:
            t, c = entry
            b2, hh, ww, _ = t.shape
            term = ops.halves_sq_sum(t, inv_h2)
        terms.append(term)
        coefs.append(1.0 / ((b2 // 2 * c * hh * ww) * len(features)))
    return ops.weighted_sum(terms, coefs)  # mean over the maps of each map's mean (loss.py:104-109)
