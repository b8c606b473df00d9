"""The side-cars of the reference's src/core/evaluation.py that the training loop calls:
``Logger`` (evaluation.py:269-308, same text byte for byte), ``model_checkpoint``
(evaluation.py:227-263, same dictionary keys and optimiser-state layout), ``image_checkpoint`` /
``save_grid`` (evaluation.py:84-221, same files and grid contents) -- plus ``load_checkpoint``,
which the reference lacks (its infinite_run.sh restarts from step 0).  FID/KID
(``val_checkpoint``) stays out of scope: clean-fid downloads its Inception weights."""

from __future__ import annotations

import math
import warnings
from pathlib import Path

import numpy as np
import torch

# (tracker attribute, label in the printed line) in the order the reference prints them
# (evaluation.py:288-304); the two accuracies share one "a/b" field.
_LINE = (
    ("log_total_disc_losses", "D loss: "),
    ("log_disc_real_accs", "D real/fake acc: "),
    ("log_disc_fake_accs", None),
    ("log_total_gen_losses", "Total G loss: "),
    ("log_gan_losses", "Gan loss "),
    ("log_idt_losses", "Idt loss "),
    ("log_rec_losses", "Rec loss "),
    ("log_kl_losses", "KL loss "),
    ("log_path_losses", "Path loss "),
    ("log_style_losses", "Style loss: "),
    ("log_ada_ps", "ADA: "),
)


class Logger:
    """Eleven lists of per-step floats; ``print`` formats their means and starts over."""

    def __init__(self, training_steps: int):
        self.training_steps = training_steps
        self.initialise_trackers()

    def initialise_trackers(self):
        for attr, _ in _LINE:
            setattr(self, attr, [])

    def print(self, step: int) -> str:
        parts = [f"Step: {step}/{self.training_steps}"]
        for attr, label in _LINE:
            mean = f"{np.mean([float(v) for v in getattr(self, attr)]):.6g}"  # (float(): LoggedScalar entries)
            if label is None:  # second half of "real/fake"
                parts[-1] += "/" + mean
            else:
                parts.append(label + mean)
        self.initialise_trackers()
        return ", ".join(parts) + ", "


# --------------------------------------------------------------------------- model checkpoint

_NETS = ("generator", "discriminator", "mapping_network", "style_extractor")


def _run_dir(config) -> Path:
    return Path(config["training"]["checkpoint_directory"]) / config["training"]["training_run"]


def model_checkpoint(step, config, generator, discriminator, mapping_network, style_extractor, generator_optimiser,
                     discriminator_optimiser, mapping_network_optimiser, style_extractor_optimiser, ada_p,
                     image_buffer) -> Path:
    """``<run>/models/<step+1>.tar`` with the reference's eleven keys (evaluation.py:246-262):
    ``<net>_state_dict``, ``<net>_optim_state_dict`` (torch.optim.Adam layout), ``ada_p``,
    ``image_buffer_images`` (list of (1,C,H,W) tensors), ``image_buffer_size``.  Two extra keys let
    ``load_checkpoint`` resume exactly: ``step`` and ``ada_state`` (the controller's open window)."""
    out_dir = _run_dir(config) / "models"
    out_dir.mkdir(parents=True, exist_ok=True)
    nets = dict(zip(_NETS, (generator, discriminator, mapping_network, style_extractor)))
    opts = dict(zip(_NETS, (generator_optimiser, discriminator_optimiser, mapping_network_optimiser,
                            style_extractor_optimiser)))
    blob = {}
    for name in _NETS:
        blob[f"{name}_state_dict"] = nets[name].state_dict()
        blob[f"{name}_optim_state_dict"] = opts[name].state_dict()
    blob["ada_p"] = ada_p()
    blob["image_buffer_images"] = [im.detach().float().contiguous().cpu() for im in image_buffer.images]
    blob["image_buffer_size"] = image_buffer.buffer_size
    blob["step"] = step + 1
    blob["ada_state"] = {"curr_batch": ada_p.curr_batch, "scores": [float(s) for s in ada_p.mean_real_scores]}
    path = out_dir / f"{step + 1}.tar"
    torch.save(blob, path)
    return path


def load_checkpoint(path, device, generator, discriminator, mapping_network, style_extractor,
                    generator_optimiser=None, discriminator_optimiser=None, mapping_network_optimiser=None,
                    style_extractor_optimiser=None, ada_p=None, image_buffer=None) -> int:
    """Resume from a checkpoint written by ``model_checkpoint`` -- or by the reference itself: the
    keys, the state-dict names / shapes and the ``torch.optim.Adam`` state layout are the same.
    (Round-1 files, which spelled ``*_optimiser_state_dict`` / ``image_buffer``, still load.)
    Anything asked for but absent from the file is reported with a warning, never skipped
    silently.  Returns the step to continue from (0 for a reference file, which records none)."""
    ck = torch.load(path, map_location=device, weights_only=True)
    nets = dict(zip(_NETS, (generator, discriminator, mapping_network, style_extractor)))
    opts = dict(zip(_NETS, (generator_optimiser, discriminator_optimiser, mapping_network_optimiser,
                            style_extractor_optimiser)))
    for name in _NETS:
        nets[name].load_state_dict(ck[f"{name}_state_dict"])
        if opts[name] is None:
            continue
        sd = ck.get(f"{name}_optim_state_dict", ck.get(f"{name}_optimiser_state_dict"))
        if sd is None:
            warnings.warn(f"{path}: no optimiser state for the {name}; its Adam moments restart from zero")
        else:
            opts[name].load_state_dict(sd)
    if ada_p is not None:
        ada_p.p = torch.tensor(float(ck.get("ada_p", 0.0)))
        st = ck.get("ada_state")
        if st:
            ada_p.curr_batch = int(st["curr_batch"])
            ada_p.mean_real_scores = [torch.tensor(s) for s in st["scores"]]
    if image_buffer is not None:
        images = ck.get("image_buffer_images", ck.get("image_buffer"))
        if images is None:
            warnings.warn(f"{path}: no image history pool; the ImageBuffer restarts empty")
        else:
            image_buffer.images = [im.to(device) for im in images]
            image_buffer.num_imgs = len(image_buffer.images)
        if "image_buffer_size" in ck and int(ck["image_buffer_size"]) != image_buffer.buffer_size:
            warnings.warn(f"{path}: pool of {ck['image_buffer_size']} images loaded into a buffer of "
                          f"{image_buffer.buffer_size}")
    return int(ck.get("step", 0))


# ---------------------------------------------------------------------------- image checkpoint


def _to_unit_range(image: torch.Tensor) -> np.ndarray:
    """(C,H,W) -> (H,W,C) stretched to [0, 1] by its own min / max (evaluation.py:92-96)."""
    im = image.detach().float().permute(1, 2, 0)
    lo, hi = im.min(), im.max()
    return ((im - lo) / (hi - lo)).cpu().numpy()


def save_grid(images, save_path, grid_size) -> None:
    """``images[col][row]`` tensors drawn as a ``grid_size = (rows, cols)`` sheet of axis-less
    panels (gray colour map for single-channel images), 300 dpi, tight box."""
    import matplotlib

    matplotlib.use("Agg", force=False)
    from matplotlib import pyplot as plt

    rows, cols = grid_size
    was_interactive = plt.isinteractive()
    plt.ioff()
    fig, axes = plt.subplots(nrows=rows, ncols=cols, figsize=(cols, rows), squeeze=False)
    for c in range(cols):
        for r in range(rows):
            ax = axes[r, c]
            ax.imshow(_to_unit_range(images[c][r]), cmap="gray")
            ax.set_axis_off()
    fig.subplots_adjust(wspace=0.1, hspace=0.1)
    fig.savefig(Path(save_path), dpi=300, bbox_inches="tight")
    plt.close(fig)
    if was_interactive:
        plt.ion()


def _first_eight(it, device, batch_size):
    need = max(1, math.ceil(8 / batch_size))
    got = torch.cat([next(it).to(device) for _ in range(need)], dim=0) if need > 1 else next(it).to(device)
    return got[:8]


def image_checkpoint(step, config, device, shoeprint_iter, shoemark_iter, mapping_network, generator,
                     style_extractor) -> tuple[Path, Path]:
    """The two sheets of evaluation.py:122-221 under ``<run>/images``:
    ``translation_<step+1>.png`` -- 8 shoeprints (top row), each decoded with the SAME 8 sampled
    styles (9 x 8); ``decoding_<step+1>.png`` -- per column: shoeprint, its zero-style
    reconstruction, its translation with a real shoemark's extracted style, that shoemark, the
    shoemark's own reconstruction (5 x 8).  Call under ``torch.no_grad()`` with the nets in eval mode
    as the reference's loop does (train.py:269-298)."""
    out_dir = _run_dir(config) / "images"
    out_dir.mkdir(parents=True, exist_ok=True)
    n_blocks = generator.n_style_blocks
    w = mapping_network.get_single_w(batch_size=8, n_gen_blocks=n_blocks, device=device, mix_styles=False,
                                     domain_variable=1)
    b = config["training"]["batch_size"]
    prints = _first_eight(shoeprint_iter, device, b)
    marks = _first_eight(shoemark_iter, device, b)
    z_print, z_mark = generator.encode(prints), generator.encode(marks)

    sheet = []
    for col in range(8):
        styled = generator.decode(z_print[col].expand(8, -1, -1, -1), w)
        sheet.append([prints[col], *styled])
    p_translation = out_dir / f"translation_{step + 1}.png"
    save_grid(sheet, p_translation, (9, 8))

    w_zero = torch.zeros((n_blocks, 8, config["architecture"]["w_dim"]), device=device)
    rec_print = generator.decode(z_print, w_zero)
    w_mark = style_extractor(marks)
    w_mark = w_mark.expand(n_blocks, *w_mark.shape)
    rec_mark = generator.decode(z_mark, w_mark)
    translated = generator.decode(z_print, w_mark)
    sheet = [[prints[c], rec_print[c], translated[c], marks[c], rec_mark[c]] for c in range(8)]
    p_decoding = out_dir / f"decoding_{step + 1}.png"
    save_grid(sheet, p_decoding, (5, 8))
    return p_translation, p_decoding
