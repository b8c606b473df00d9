#!/usr/bin/env python3
"""Training loop on the MI355X hot path: the body of the reference's train.py:28-319 with
its own ``config.toml`` (same keys), minus the parts that are out of scope (FID/KID,
matplotlib grids, the author's image folders).

  python train.py config.toml [--steps N] [--synthetic] [--resume ckpt.tar] [--precision bf16|fp32]

Data: ``--synthetic`` (default when the configured directories do not exist) draws
uniform [-1, 1) images resident in HBM, as bench.py does.  Real data: image folders through the
device-resident pool of one_to_many_gan_amd/data/datasets.py, or any iterator of
(B, C, H, W) tensors can be handed to ``run``.
"""

from __future__ import annotations

import argparse
import itertools
import random
import sys
import time

import numpy as np
import torch

import one_to_many_gan_amd as o2m
from one_to_many_gan_amd.core.evaluation import Logger, load_checkpoint, model_checkpoint
from one_to_many_gan_amd.core.training import ImageBuffer, discriminator_step, generator_step
from one_to_many_gan_amd.data.config import load_config
from one_to_many_gan_amd.model.builder import Discriminator, Generator, MappingNetwork, StyleExtractor
from one_to_many_gan_amd.model.loss import ADAp


def synthetic_batches(seed, config, device, n_distinct=8):
    g = torch.Generator().manual_seed(seed)
    b, c = config["training"]["batch_size"], config["data"]["image_channels"]
    h, w = config["data"]["image_size"]
    pool = [(torch.rand(b, c, h, w, generator=g) * 2 - 1).to(device) for _ in range(n_distinct)]
    return itertools.cycle(pool)


def build(config, device):
    """Seeds, the four networks and their (fused) Adam optimisers: train.py:35-116."""
    seed = config["training"]["random_seed"]
    torch.manual_seed(seed)
    np.random.default_rng(seed)
    random.seed(seed)
    torch.cuda.manual_seed_all(seed)
    a, d, t, o = config["architecture"], config["data"], config["training"], config["optimisation"]
    nets = {
        "D": Discriminator(input_nc=d["image_channels"]).to(device),
        "G": Generator(input_nc=d["image_channels"], w_dim=a["w_dim"], image_size=d["image_size"],
                       min_latent_resolution=a["min_latent_resolution"],
                       n_resnet_blocks=a["n_resnet_blocks"]).to(device),
        "M": MappingNetwork(features=a["w_dim"], n_layers=a["mapping_network_layers"],
                            style_mixing_prob=t["style_mixing_prob"]).to(device),
        "S": StyleExtractor(input_nc=d["image_channels"], w_dim=a["w_dim"]).to(device),
    }
    betas = tuple(o["adam_betas"])
    opts = {k: o2m.make_adam(n, o["mapping_network_learning_rate"] if k == "M" else o["learning_rate"], betas)
            for k, n in nets.items()}
    return nets, opts


def run(config, device, steps, shoeprint_iter, shoemark_iter, resume=None, log=print):
    nets, opts = build(config, device)
    image_buffer = ImageBuffer(config["training"]["image_buffer_size"])
    ada = o2m.AdaptiveDiscriminatorAugmentation(**o2m.REFERENCE_ADA_SWITCHES).to(device)  # train.py:175-188
    ada_p = ADAp(ada_e=config["ada"]["ada_overfitting_measurement_n_images"],
                 ada_adjustment_size=config["ada"]["ada_adjustment_size"],
                 batch_size=config["training"]["batch_size"],
                 discriminator_overfitting_target=config["ada"]["discriminator_real_acc_target"])
    first = 0
    if resume:
        first = load_checkpoint(resume, device, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["D"],
                                opts["M"], opts["S"], ada_p, image_buffer)
        log(f"resumed from {resume} at step {first}")
    logger = Logger(steps)
    ev = config["evaluation"]
    t0 = time.perf_counter()
    for step in range(first, steps):
        p = ada_p()
        ada.set_p(p)
        logger.log_ada_ps.append(p)
        d_loss, (real_acc, fake_acc) = discriminator_step(
            config, device, nets["D"], nets["G"], nets["M"], opts["D"], shoeprint_iter, shoemark_iter,
            image_buffer, ada, ada_p)
        logger.log_total_disc_losses.append(d_loss)
        logger.log_disc_real_accs.append(real_acc)
        logger.log_disc_fake_accs.append(fake_acc)
        g_loss, (gan, rec, idt, kl, path, style) = generator_step(
            config, device, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["M"], opts["S"],
            shoeprint_iter, shoemark_iter, ada)
        for name, v in (("total_gen", g_loss), ("gan", gan), ("rec", rec), ("idt", idt), ("kl", kl),
                        ("path", path), ("style", style)):
            getattr(logger, f"log_{name}_losses").append(v)
        if (step + 1) % ev["log_interval"] == 0 or step + 1 == steps:
            dt = time.perf_counter() - t0
            log(logger.print(step + 1) + f" | {dt:.1f}s")
        if (step + 1) % ev["checkpoint_interval"] == 0 or step + 1 == steps:
            path_ = model_checkpoint(step, config, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["D"],
                                     opts["M"], opts["S"], ada_p, image_buffer)
            log(f"checkpoint {path_}")
    return nets, opts


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?", default="config.toml")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--resume", default=None)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    args = ap.parse_args(argv)
    config = load_config(args.config)
    if not torch.cuda.is_available():
        sys.exit("train.py drives the MI355X hot path: no GPU visible (there is no CPU fallback)")
    device = torch.device(f"cuda:{config['training']['gpu_number']}")
    o2m.set_precision(args.precision)
    have_data = config["data"]["shoeprint_data_dir"].exists() and config["data"]["shoemark_data_dir"].exists()
    steps = args.steps if args.steps is not None else config["training"]["training_steps"]
    if args.synthetic or not have_data:
        prints, marks = synthetic_batches(1000, config, device), synthetic_batches(2000, config, device)
    else:
        # reference train.py:118-169 with the images resident in HBM (data/datasets.py)
        from one_to_many_gan_amd.data import datasets as D

        tf = D.Compose([D.Resize(tuple(config["data"]["image_size"])), D.ToTensor(), D.Normalize((0.5,), (0.5,))])
        g = torch.Generator().manual_seed(config["training"]["random_seed"])
        loaders = []
        for key in ("shoeprint_data_dir", "shoemark_data_dir"):
            pool = D.DeviceImagePool(D.ShoeDataset(config["data"][key], mode="train", transform=tf), device)
            loaders.append(D.DeviceLoader(pool, config["training"]["batch_size"], generator=g))
        prints, marks = loaders[0].cycle(), loaders[1].cycle()
    run(config, device, steps, prints, marks, resume=args.resume)


if __name__ == "__main__":
    main()
