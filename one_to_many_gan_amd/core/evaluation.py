"""The per-step side-cars of the reference's src/core/evaluation.py that belong to the
training loop: ``Logger`` (evaluation.py:269-308) and ``model_checkpoint``
(evaluation.py:227-263) -- plus ``load_checkpoint``, which the reference lacks (its
infinite_run.sh restarts from step 0).  FID/KID and the matplotlib image grids are out of
scope (they need network-fetched Inception weights / torchvision)."""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch


class Logger:
    """Accumulates the 11 per-step scalars and prints their means, then resets."""

    _FIELDS = ("log_total_gen_losses", "log_gan_losses", "log_rec_losses", "log_idt_losses", "log_kl_losses",
               "log_path_losses", "log_style_losses", "log_total_disc_losses", "log_disc_real_accs",
               "log_disc_fake_accs", "log_ada_ps")

    def __init__(self, training_steps: int):
        self.training_steps = training_steps
        self._reset()

    def _reset(self):
        for f in self._FIELDS:
            setattr(self, f, [])

    def print(self, step: int) -> str:
        m = {f: float(np.mean(getattr(self, f))) if getattr(self, f) else float("nan") for f in self._FIELDS}
        line = (f"[{step}/{self.training_steps}] "
                f"G {m['log_total_gen_losses']:.4f} (gan {m['log_gan_losses']:.4f} rec {m['log_rec_losses']:.4f} "
                f"idt {m['log_idt_losses']:.4f} kl {m['log_kl_losses']:.4f} path {m['log_path_losses']:.4f} "
                f"style {m['log_style_losses']:.4f}) | D {m['log_total_disc_losses']:.4f} "
                f"(real {m['log_disc_real_accs']:.3f} fake {m['log_disc_fake_accs']:.3f}) | "
                f"ada p {m['log_ada_ps']:.4f}")
        self._reset()
        return line


def _opt_state(opt):
    return opt.state_dict()


def model_checkpoint(step, config, generator, discriminator, mapping_network, style_extractor, generator_optimiser,
                     discriminator_optimiser, mapping_network_optimiser, style_extractor_optimiser, ada_p,
                     image_buffer) -> Path:
    """Same dictionary keys as the reference's ``<step+1>.tar`` (evaluation.py:238-262)."""
    out_dir = Path(config["training"]["checkpoint_directory"]) / config["training"]["training_run"] / "models"
    out_dir.mkdir(parents=True, exist_ok=True)
    path = out_dir / f"{step + 1}.tar"
    torch.save({
        "step": step + 1,
        "generator_state_dict": generator.state_dict(),
        "discriminator_state_dict": discriminator.state_dict(),
        "mapping_network_state_dict": mapping_network.state_dict(),
        "style_extractor_state_dict": style_extractor.state_dict(),
        "generator_optimiser_state_dict": _opt_state(generator_optimiser),
        "discriminator_optimiser_state_dict": _opt_state(discriminator_optimiser),
        "mapping_network_optimiser_state_dict": _opt_state(mapping_network_optimiser),
        "style_extractor_optimiser_state_dict": _opt_state(style_extractor_optimiser),
        "ada_p": ada_p(),
        "ada_state": {"curr_batch": ada_p.curr_batch,
                      "scores": [float(s) for s in ada_p.mean_real_scores]},
        "image_buffer": [im.float().cpu() for im in image_buffer.images],
        "image_buffer_size": image_buffer.buffer_size,
    }, path)
    return path


def load_checkpoint(path, device, generator, discriminator, mapping_network, style_extractor,
                    generator_optimiser=None, discriminator_optimiser=None, mapping_network_optimiser=None,
                    style_extractor_optimiser=None, ada_p=None, image_buffer=None) -> int:
    """Resume from a checkpoint written by ``model_checkpoint`` -- or by the reference itself:
    the four ``*_state_dict`` entries have identical keys and shapes.  Returns the step."""
    ck = torch.load(path, map_location=device, weights_only=True)
    generator.load_state_dict(ck["generator_state_dict"])
    discriminator.load_state_dict(ck["discriminator_state_dict"])
    mapping_network.load_state_dict(ck["mapping_network_state_dict"])
    style_extractor.load_state_dict(ck["style_extractor_state_dict"])
    for opt, key in ((generator_optimiser, "generator"), (discriminator_optimiser, "discriminator"),
                     (mapping_network_optimiser, "mapping_network"), (style_extractor_optimiser, "style_extractor")):
        sd = ck.get(f"{key}_optimiser_state_dict")
        if opt is not None and sd is not None and "exp_avg" in sd:  # fused-Adam layout only
            opt.load_state_dict(sd)
    if ada_p is not None:
        ada_p.p = torch.tensor(float(ck.get("ada_p", 0.0)))
        st = ck.get("ada_state")
        if st:
            ada_p.curr_batch = int(st["curr_batch"])
            ada_p.mean_real_scores = [torch.tensor(s) for s in st["scores"]]
    if image_buffer is not None and "image_buffer" in ck:
        image_buffer.images = [im.to(device) for im in ck["image_buffer"]]
        image_buffer.num_imgs = len(image_buffer.images)
    return int(ck.get("step", 0))
