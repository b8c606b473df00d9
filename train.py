#!/usr/bin/env python3
"""Training loop on the MI355X hot path: the body of the reference's train.py:28-319 with
its own ``config.toml`` (same keys), minus the parts that are out of scope (FID/KID,
matplotlib grids, the author's image folders).

  python train.py config.toml [--steps N] [--synthetic] [--resume ckpt.tar] [--precision bf16|fp32]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 train.py config.toml
      (data parallel: one process per GPU, RCCL gradient all-reduce, batch_size per GPU)

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
import os
import time
from pathlib import Path

import numpy as np
import torch

import one_to_many_gan_amd as o2m
from one_to_many_gan_amd.core.evaluation import Logger, image_checkpoint, load_checkpoint, model_checkpoint
from one_to_many_gan_amd.core.training import ImageBuffer, discriminator_step, generator_step, set_async_scalars
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


def run(config, device, steps, shoeprint_iter, shoemark_iter, resume=None, log=print, data_parallel=False,
        image_grids=True, graph=False):
    """The loop of the reference's train.py:171-319.  ``data_parallel``: torch.distributed is
    initialised (one rank per GPU); gradients, the KL moments and the ADA confidence are reduced
    over ranks (one_to_many_gan_amd/dist.py) and only rank 0 logs and writes checkpoints.
    ``graph``: the D+G step as one replayed HIP graph (one_to_many_gan_amd/core/graphed.py) -- for shapes whose step
    is bound by the host (a 64x64 step: 14.5 -> 7.8 ms); single process, augmentation held at the identity."""
    if graph:
        return _run_graphed(config, device, steps, shoeprint_iter, shoemark_iter, resume, log, image_grids)
    if device.type == "cuda":
        # the launchers enqueue on the CURRENT stream of the tensors' device; make that device the
        # process's current one as well so allocations and events follow (train.py:61-65)
        torch.cuda.set_device(device)
    o2m.ops.set_deterministic(bool(config["training"].get("deterministic_cuda_kernels", False)))
    nets, opts = build(config, device)
    image_buffer = ImageBuffer(config["training"]["image_buffer_size"])
    ada = o2m.AdaptiveDiscriminatorAugmentation(**o2m.REFERENCE_ADA_SWITCHES).to(device)  # train.py:175-188
    ada_p = ADAp(ada_e=config["ada"]["ada_overfitting_measurement_n_images"],
                 ada_adjustment_size=config["ada"]["ada_adjustment_size"],
                 batch_size=config["training"]["batch_size"],
                 discriminator_overfitting_target=config["ada"]["discriminator_real_acc_target"])
    kl_hook, rank = None, 0
    if data_parallel:
        import torch.distributed as dist

        from one_to_many_gan_amd import dist as o2m_dist

        rank = dist.get_rank()
        o2m_dist.broadcast_parameters(opts.values())
        for o in opts.values():
            o2m_dist.BucketReducer(o)
        kl_hook = o2m_dist.make_kl_moment_hook()
        o2m_dist.sync_ada_p(ada_p)
        torch.manual_seed(config["training"]["random_seed"] + 1 + rank)  # z / theta / h differ per rank
    first = 0
    if resume:
        first = load_checkpoint(resume, device, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["D"],
                                opts["M"], opts["S"], ada_p, image_buffer)
        log(f"resumed from {resume} at step {first}")
    logger = Logger(steps)
    # the ten logged scalars of a step reach the logger's lists as LoggedScalar objects and are read when a line is
    # printed: no blocking read per step (the reference's .item() calls, training.py:125-128,250-257)
    set_async_scalars(device.type == "cuda")
    ev = config["evaluation"]
    t0 = time.perf_counter()
    throttle = o2m.ops.StepThrottle(device)  # at most two steps queued on the device (the host issues them ~2.5x faster)
    for step in range(first, steps):
        p = ada_p()
        ada.set_p(p)
        logger.log_ada_ps.append(p)
        throttle.__enter__()
        d_loss, (real_acc, fake_acc) = discriminator_step(
            config, device, nets["D"], nets["G"], nets["M"], opts["D"], shoeprint_iter, shoemark_iter,
            image_buffer, ada, ada_p)
        logger.log_total_disc_losses.append(d_loss)
        logger.log_disc_real_accs.append(real_acc)
        logger.log_disc_fake_accs.append(fake_acc)
        g_loss, (gan, rec, idt, kl, path, style) = generator_step(
            config, device, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["M"], opts["S"],
            shoeprint_iter, shoemark_iter, ada, kl_moment_hook=kl_hook)
        throttle.__exit__(None, None, None)
        for name, v in (("total_gen", g_loss), ("gan", gan), ("rec", rec), ("idt", idt), ("kl", kl),
                        ("path", path), ("style", style)):
            getattr(logger, f"log_{name}_losses").append(v)
        if rank != 0:
            if (step + 1) % ev["log_interval"] == 0:
                logger.initialise_trackers()  # only rank 0 prints (and thereby resets) the windows
            continue
        if (step + 1) % ev["log_interval"] == 0 or step + 1 == steps:
            line = logger.print(step + 1)  # the reference's line, also appended to <run>/log (train.py:253-267)
            log(line)
            run_dir = Path(config["training"]["checkpoint_directory"]) / config["training"]["training_run"]
            run_dir.mkdir(parents=True, exist_ok=True)
            with (run_dir / "log").open("a") as f:
                f.write(line + "\n")
            log(f"elapsed {time.perf_counter() - t0:.1f}s")
        if (step + 1) % ev["checkpoint_interval"] == 0 or step + 1 == steps:
            # train.py:269-315 without val_checkpoint (FID / KID need network-fetched weights)
            for k in ("G", "M", "S"):
                nets[k].eval()
            if image_grids:
                with torch.no_grad():
                    grids = image_checkpoint(step, config, device, shoeprint_iter, shoemark_iter, nets["M"],
                                             nets["G"], nets["S"])
                log(f"image grids {grids[0]} {grids[1]}")
            path_ = model_checkpoint(step, config, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["D"],
                                     opts["M"], opts["S"], ada_p, image_buffer)
            log(f"checkpoint {path_}")
            for k in ("G", "M", "S"):
                nets[k].train()
    return nets, opts


def _run_graphed(config, device, steps, shoeprint_iter, shoemark_iter, resume, log, image_grids):
    """``run(graph=True)``: same networks, optimisers, log line and checkpoint files; the step itself is
    core.graphed.GraphedStep (device-resident style draws, history pool, ADAp controller and scalar sums)."""
    from one_to_many_gan_amd.core.graphed import GraphedStep

    if device.type != "cuda":
        raise RuntimeError("the graphed loop needs a GPU")
    torch.cuda.set_device(device)
    o2m.ops.set_deterministic(bool(config["training"].get("deterministic_cuda_kernels", False)))
    nets, opts = build(config, device)
    gs = GraphedStep(config, device, nets, opts, shoeprint_iter, shoemark_iter, o2m.IdentityADA())
    log("graphed step: augmentation held at the identity; style draws, history pool and ADAp on the device")
    first, p_held = 0, 0.0
    if resume:
        ref_p = ADAp(1, 0.0, 1, config["ada"]["discriminator_real_acc_target"])
        ref_buf = ImageBuffer(config["training"]["image_buffer_size"])
        first = load_checkpoint(resume, device, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["D"],
                                opts["M"], opts["S"], ref_p, ref_buf)
        gs.ada_p.load_reference(ref_p)
        gs.buffer.load_reference(ref_buf, device)
        p_held = float(ref_p.p)
        log(f"resumed from {resume} at step {first}")
    # The device controller keeps integrating the discriminator's confidence, but no augmentation feeds back: its p only
    # drifts up (1.18 after 600 steps, profiles/r03_long_run.txt).  That drift is NOT what this run trained with: the log
    # line and the checkpoints carry the p the run started from (an eager resume, or the reference itself, would otherwise
    # start augmenting at the drifted value).
    log(f"graphed step: ADA probability held at {p_held:g} in the log and the checkpoints (the controller's own value is not used)")
    logger = Logger(steps)
    ev = config["evaluation"]
    t0 = time.perf_counter()
    for step in range(first, steps):
        gs.step()
        last = step + 1 == steps
        if (step + 1) % ev["log_interval"] == 0 or last:
            (d_loss, real_acc, fake_acc), g_vals = gs.logged_means()  # window means, summed on the device
            logger.log_ada_ps.append(p_held)
            logger.log_total_disc_losses.append(d_loss)
            logger.log_disc_real_accs.append(real_acc)
            logger.log_disc_fake_accs.append(fake_acc)
            for name, v in zip(("total_gen", "gan", "rec", "idt", "kl", "path", "style"), g_vals):
                getattr(logger, f"log_{name}_losses").append(v)
            line = logger.print(step + 1)
            log(line)
            run_dir = Path(config["training"]["checkpoint_directory"]) / config["training"]["training_run"]
            run_dir.mkdir(parents=True, exist_ok=True)
            with (run_dir / "log").open("a") as f:
                f.write(line + "\n")
            log(f"elapsed {time.perf_counter() - t0:.1f}s" + (" (graph replay)" if gs.graph is not None else ""))
        if (step + 1) % ev["checkpoint_interval"] == 0 or last:
            for k in ("G", "M", "S"):
                nets[k].eval()
            if image_grids:
                draws = nets["M"].device_draws
                nets["M"].device_draws = False  # the evaluation grids use the reference's CPU draws
                with torch.no_grad():
                    grids = image_checkpoint(step, config, device, shoeprint_iter, shoemark_iter, nets["M"],
                                             nets["G"], nets["S"])
                nets["M"].device_draws = draws
                log(f"image grids {grids[0]} {grids[1]}")
            ref_p_out = gs.ada_p.to_reference()
            ref_p_out.p = torch.tensor(p_held)
            path_ = model_checkpoint(step, config, nets["G"], nets["D"], nets["M"], nets["S"], opts["G"], opts["D"],
                                     opts["M"], opts["S"], ref_p_out, gs.buffer.to_reference())
            log(f"checkpoint {path_}")
            for k in ("G", "M", "S"):
                nets[k].train()
    return nets, opts


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?", default="config.toml")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--resume", default=None)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", action="store_true",
                    help="replay the D+G step as one HIP graph (host-bound shapes; single GPU, augmentation off)")
    args = ap.parse_args(argv)
    config = load_config(args.config)
    if not torch.cuda.is_available():
        sys.exit("train.py drives the MI355X hot path: no GPU visible (there is no CPU fallback)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:  # one rank per GPU; gpu_number names the single-process device only
        import torch.distributed as dist

        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(device)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("O2M_DIST_BACKEND", "nccl"),
                                **({"device_id": device} if os.environ.get("O2M_DIST_BACKEND", "nccl") == "nccl" else {}))
    else:
        device = torch.device(f"cuda:{config['training']['gpu_number']}")
        torch.cuda.set_device(device)
    o2m.set_precision(args.precision)
    have_data = config["data"]["shoeprint_data_dir"].exists() and config["data"]["shoemark_data_dir"].exists()
    steps = args.steps if args.steps is not None else config["training"]["training_steps"]
    rank = int(os.environ.get("RANK", "0"))
    if args.synthetic or not have_data:
        prints, marks = synthetic_batches(1000 + rank, config, device), synthetic_batches(2000 + rank, config, device)
    else:
        # reference train.py:118-169 with the images resident in HBM (data/datasets.py)
        from one_to_many_gan_amd.data import datasets as D

        tf = D.Compose([D.Resize(tuple(config["data"]["image_size"])), D.ToTensor(), D.Normalize((0.5,), (0.5,))])
        g = torch.Generator().manual_seed(config["training"]["random_seed"] + rank)
        loaders = []
        for key in ("shoeprint_data_dir", "shoemark_data_dir"):
            pool = D.DeviceImagePool(D.ShoeDataset(config["data"][key], mode="train", transform=tf), device)
            loaders.append(D.DeviceLoader(pool, config["training"]["batch_size"], generator=g))
        prints, marks = loaders[0].cycle(), loaders[1].cycle()
    if args.graph and world > 1:
        sys.exit("--graph is single-process")
    run(config, device, steps, prints, marks, resume=args.resume, data_parallel=world > 1, graph=args.graph)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
