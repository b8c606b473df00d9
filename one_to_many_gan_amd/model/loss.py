"""Losses and the ADA probability controller with the reference's signatures
(src/model/loss.py).  The image/feature-sized reductions (KL moments, path energy) run as
HIP reduction kernels on the NHWC buffers; B x w_dim arithmetic stays in torch."""

from __future__ import annotations

import torch
from torch.nn import functional as F

from .. import ops


class ADAp:
    """Adaptive-discriminator-augmentation probability (reference loss.py:11-52): a host
    state machine fed one 0-dim score per discriminator step.  Scores are kept as device
    tensors; nothing here forces a device sync until ``__call__`` is asked for the float."""

    def __init__(self, ada_e: float, ada_adjustment_size: float, batch_size: int,
                 discriminator_overfitting_target: float):
        self.n_batches = ada_e // batch_size
        self.ada_adjustment = ada_adjustment_size * ada_e
        self.overfitting_target = discriminator_overfitting_target
        self.p = torch.zeros(())
        self.curr_batch = 0
        self.mean_real_scores = []

    def update_p(self, mean_score: torch.Tensor):
        window_closes = self.curr_batch == self.n_batches
        self.mean_real_scores.append(mean_score)
        if window_closes:
            mean_sign = torch.stack([s.detach().float().cpu() for s in self.mean_real_scores]).mean()
            if mean_sign < self.overfitting_target:
                self.p = self.p - self.ada_adjustment
            elif mean_sign > self.overfitting_target:
                self.p = self.p + self.ada_adjustment
            self.p = torch.clamp_min(self.p, 0.0)
            # the score that closed the window also opens the next one (reference loss.py:34,49)
            self.mean_real_scores = [mean_score]
            self.curr_batch = 0
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
