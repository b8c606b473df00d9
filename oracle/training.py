"""Oracle losses and step functions (TEST ONLY): restates src/model/loss.py and
src/core/training.py of the reference in plain torch fp32.

The ``ada`` argument is the augmentation slot (pytorch-ada in the reference, not available
offline).  Every parity configuration holds it at p = 0, i.e. identity; ``IdentityADA`` is
that stand-in ("parity unpinned" for the augmentation arithmetic itself).
"""

from __future__ import annotations

import random

import torch
import torch.nn.functional as F


class IdentityADA(torch.nn.Module):
    """Interface slot of ada.AdaptiveDiscriminatorAugmentation (train.py:175-188,206)."""

    def __init__(self, **_kw):
        super().__init__()
        self.p = 0.0

    def set_p(self, p):
        self.p = float(p)

    def forward(self, x):
        return x


class ADAp:
    """loss.py:11-52 -- augmentation-probability controller (host state machine)."""

    def __init__(self, ada_e, ada_adjustment_size, batch_size, discriminator_overfitting_target):
        self.n_batches = ada_e // batch_size
        self.ada_adjustment = ada_adjustment_size * ada_e
        self.overfitting_target = discriminator_overfitting_target
        self.p = torch.zeros(())
        self.curr_batch = 0
        self.mean_real_scores = []

    def update_p(self, mean_score):
        if self.curr_batch == self.n_batches:
            # the triggering score closes this window and (below) opens the next (loss.py:34,49)
            self.mean_real_scores.append(mean_score)
            m = torch.stack(self.mean_real_scores).mean()
            if m < self.overfitting_target:
                self.p = self.p - self.ada_adjustment
            elif m > self.overfitting_target:
                self.p = self.p + self.ada_adjustment
            self.curr_batch = 0
            self.mean_real_scores = []
            self.p = torch.relu(self.p)
        self.curr_batch += 1
        self.mean_real_scores.append(mean_score)

    def __call__(self):
        return self.p.item()


class ImageBuffer:
    """training.py:22-65 -- CycleGAN history pool driven by Python ``random``."""

    def __init__(self, buffer_size):
        if buffer_size < 1:
            raise ValueError
        self.buffer_size = buffer_size
        self.num_imgs = 0
        self.images = []

    def __call__(self, images):
        out = []
        for img in images:
            img = img.detach().unsqueeze(0)
            if self.num_imgs < self.buffer_size:
                self.num_imgs += 1
                self.images.append(img)
                out.append(img)
            elif random.uniform(0, 1) > 0.5:
                j = random.randint(0, self.buffer_size - 1)
                out.append(self.images[j].clone())
                self.images[j] = img
            else:
                out.append(img)
        return torch.cat(out, 0)


def style_cycle_loss_func(original_w, reconstructed_w, *, normalise=True, cos_l2_ratio=0.2):
    """loss.py:60-75."""
    if normalise:
        original_w = F.normalize(original_w, dim=-1)
        reconstructed_w = F.normalize(reconstructed_w, dim=-1)
    cos = F.cosine_similarity(original_w, reconstructed_w, dim=-1).mean()
    return (1 - cos) + cos_l2_ratio * F.mse_loss(original_w, reconstructed_w)


def kl_loss_func(combined_latents):
    """loss.py:82-92 -- global mean^2 + (biased var - 1)^2."""
    m = combined_latents.mean()
    v = combined_latents.var(correction=0)
    return m * m + (v - 1) ** 2


def path_loss_func(features1, features2, cent_fin_diff_h):
    """loss.py:98-111."""
    total = torch.zeros((), device=features1[0].device)
    for a, b in zip(features1, features2, strict=True):
        total = total + (((a - b) / cent_fin_diff_h[:, None, None, None]) ** 2).mean()
    return total / len(features1)


def _f(x):
    return x.detach().cpu().item()


def discriminator_step(config, device, discriminator, generator, mapping_network,
                       discriminator_optimiser, shoeprint_iter, shoemark_iter, image_buffer,
                       ada, ada_p):
    """training.py:71-128."""
    bs = config["training"]["batch_size"]
    discriminator_optimiser.zero_grad()
    prints = next(shoeprint_iter).to(device)
    w = mapping_network.get_single_w(bs, generator.n_style_blocks, device, 1)
    fake = ada(image_buffer(generator(prints, w)))
    real = ada(next(shoemark_iter).to(device))
    fake_scores = discriminator(fake)
    real_scores = discriminator(real)
    loss = (F.mse_loss(real_scores, torch.ones_like(real_scores))
            + F.mse_loss(fake_scores, torch.zeros_like(fake_scores))) / 2
    conf = lambda s: torch.sign(s * 2 - 1).mean()  # noqa: E731  (training.py:86)
    sign_real = conf(real_scores.detach())
    sign_fake = -conf(fake_scores.detach())
    ada_p.update_p(sign_real)
    loss.backward()
    discriminator_optimiser.step()
    return _f(loss), (_f(sign_real), _f(sign_fake))


def generator_step(config, device, generator, discriminator, mapping_network, style_extractor,
                   generator_optimiser, mapping_network_optimiser, style_extractor_optimiser,
                   shoeprint_iter, shoemark_iter, ada):
    """training.py:136-257."""
    bs = config["training"]["batch_size"]
    opt = config["optimisation"]
    nb = generator.n_style_blocks
    generator_optimiser.zero_grad()
    mapping_network_optimiser.zero_grad()
    style_extractor_optimiser.zero_grad()
    prints = next(shoeprint_iter).to(device)
    marks = next(shoemark_iter).to(device)

    latents = generator.encode(torch.cat([prints, marks], 0))
    kl = kl_loss_func(latents)
    if config["architecture"]["add_latent_noise"]:
        latents = latents + torch.randn_like(latents)
    z_print, z_mark = latents.chunk(2, 0)

    rec = F.l1_loss(generator.decode(z_print, mapping_network.get_single_w(bs, nb, device, 0)), prints)

    w_mark = style_extractor(marks)
    idt = F.l1_loss(generator.decode(z_mark, w_mark.expand(nb, *w_mark.shape)), marks)

    w_t = mapping_network.get_single_w(bs, nb, device, 1)
    generated = generator.decode(z_print, w_t)
    scores = discriminator(ada(generated))
    gan = F.mse_loss(scores, torch.ones_like(scores))

    style = style_cycle_loss_func(w_t[-1], style_extractor(generated))

    theta = torch.rand(bs).to(device)
    lo, hi = opt["path_loss_jacobian_granularity"]
    h = torch.ones_like(theta).uniform_(lo, hi)
    d1 = (theta + h / 2).clamp(0, 1)
    d2 = (theta - h / 2).clamp(0, 1)
    w1, w2 = mapping_network.get_two_w(bs, nb, device, (d1, d2))
    path = path_loss_func(generator.extract(z_print, w1), generator.extract(z_print, w2), h)

    total = (gan + opt["identity_loss_lambda"] * idt + opt["reconstruction_loss_lambda"] * rec
             + opt["kl_loss_lambda"] * kl + opt["path_loss_lambda"] * path
             + opt["style_cycle_loss_lambda"] * style)
    total.backward()
    generator_optimiser.step()
    mapping_network_optimiser.step()
    style_extractor_optimiser.step()
    return _f(total), (_f(gan), _f(rec), _f(idt), _f(kl), _f(path), _f(style))
