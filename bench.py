#!/usr/bin/env python3
"""Benchmark of the one-to-many GAN hot path: one step = discriminator_step +
generator_step on synthetic 256x256 RGB batches (BASELINE.json configs[1]: B = 16 per GPU,
bf16), weak-scaled data parallel over N GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description): images/sec over the whole
job, plus `roofline` (dominant kernel, live HIP-event timing, achieved fraction of the dense
bf16 MFMA peak) and, at N = 1, `cpu_baseline` (the CPU oracle timed on the host cores on a
bounded sample of the same workload).
"""

from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_IMAGE_STEP = {(256, 3): 1461.40e9, (512, 3): 6762.64e9, (64, 1): 60.19e9}
# dense bf16 MFMA; fp32 mode issues 3 MFMAs/product; the fp8 mode's igemm kernel uses the block-scaled K = 128 MFMA
# (twice the bf16 rate on gfx950): 5 PF denominator -- harsh on the whole step, whose weight-gradient and halo-tile
# kernels still compute in bf16
MFMA_PEAK = {"bf16": 2.5e15, "fp32": 2.5e15 / 3, "fp8": 5.0e15}


def make_config(size, channels, batch):
    return {
        "training": {"batch_size": batch, "random_seed": 42, "training_steps": 150000,
                     "image_buffer_size": 100, "style_mixing_prob": 0.9,
                     "deterministic_cuda_kernels": False, "gpu_number": 0},
        "optimisation": {"style_cycle_loss_lambda": 5.0, "identity_loss_lambda": 5.0,
                         "reconstruction_loss_lambda": 5.0, "kl_loss_lambda": 0.01,
                         "path_loss_lambda": 0.1, "path_loss_jacobian_granularity": [0.1, 0.2],
                         "learning_rate": 2e-3, "mapping_network_learning_rate": 2e-5,
                         "adam_betas": [0.5, 0.99]},
        "ada": {"discriminator_real_acc_target": 0.6, "ada_overfitting_measurement_n_images": 256,
                "ada_adjustment_size": 5.12e-4},
        "architecture": {"w_dim": 6, "add_latent_noise": False, "min_latent_resolution": 64,
                         "n_resnet_blocks": 7, "mapping_network_layers": 2},
        "data": {"image_size": [size, size], "image_channels": channels},
    }


def synthetic_stream(seed, batch, channels, size, device, n_distinct=4):
    """Endless stream of image batches in [-1,1), pre-staged on `device` (no DataLoader)."""
    g = torch.Generator().manual_seed(seed)
    pool = [(torch.rand(batch, channels, size, size, generator=g) * 2 - 1).to(device) for _ in range(n_distinct)]
    i = 0
    while True:
        yield pool[i % n_distinct]
        i += 1


class Trainer:
    """Everything train.py:72-116,171-195 builds, for either code base."""

    def __init__(self, ns, cfg, device, seed_offset=0):
        a, t, d, o = cfg["architecture"], cfg["training"], cfg["data"], cfg["optimisation"]
        torch.manual_seed(t["random_seed"])
        self.ns, self.cfg, self.device = ns, cfg, device
        self.D = ns.Discriminator(d["image_channels"]).to(device)
        self.G = ns.Generator(d["image_channels"], a["w_dim"], tuple(d["image_size"]),
                              a["min_latent_resolution"], a["n_resnet_blocks"]).to(device)
        self.M = ns.MappingNetwork(a["w_dim"], a["mapping_network_layers"], t["style_mixing_prob"]).to(device)
        self.S = ns.StyleExtractor(d["image_channels"], a["w_dim"]).to(device)
        betas = tuple(o["adam_betas"])
        self.oD = ns.make_adam(self.D, o["learning_rate"], betas)
        self.oG = ns.make_adam(self.G, o["learning_rate"], betas)
        self.oM = ns.make_adam(self.M, o["mapping_network_learning_rate"], betas)
        self.oS = ns.make_adam(self.S, o["learning_rate"], betas)
        b = t["batch_size"]
        self.prints = synthetic_stream(1000 + seed_offset, b, d["image_channels"], d["image_size"][0], device)
        self.marks = synthetic_stream(2000 + seed_offset, b, d["image_channels"], d["image_size"][0], device)
        self.buffer = ns.ImageBuffer(t["image_buffer_size"])
        self.ada = ns.make_ada().to(device)
        self.ada_p = ns.ADAp(cfg["ada"]["ada_overfitting_measurement_n_images"], cfg["ada"]["ada_adjustment_size"],
                             b, cfg["ada"]["discriminator_real_acc_target"])
        self.kl_hook = None
        self.fixed_ada_p = 0.0  # --ada-p: a fixed augmentation probability (not part of the headline workload)
        torch.manual_seed(t["random_seed"] + 1 + seed_offset)  # z / theta / h streams differ per rank

    def step(self):
        # O2M_MAIN_PRIO=1 (experiment): the step's main stream is a high-priority stream, so that its kernels are
        # dispatched ahead of the side streams' (weight gradient, discriminator step, extraction group) when CUs free up
        if os.environ.get("O2M_MAIN_PRIO", "0") == "1" and self.device.type == "cuda":
            if getattr(self, "_prio", None) is None:
                self._prio = torch.cuda.Stream(device=self.device, priority=-1)
                self._prio.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._prio):
                return self._step()
        return self._step()

    def _step(self):
        """The loop body of train.py:204-251, at most two steps queued (ops.StepThrottle, as train.py runs it)."""
        th = self.__dict__.get("_throttle")
        if th is None:
            th = self._throttle = self.ns.make_throttle(self.device)
        with th:
            return self._step_body()

    def _step_body(self):
        # The benchmark workload holds the augmentation at p = 0 (the CPU oracle
        # has no transforms): the ADAp controller still runs inside discriminator_step, its output
        # is read like train.py:206 does, but not applied.
        self.ada_p()
        self.ada.set_p(self.fixed_ada_p)
        d_out = self.ns.discriminator_step(self.cfg, self.device, self.D, self.G, self.M, self.oD,
                                           self.prints, self.marks, self.buffer, self.ada, self.ada_p)
        kw = {"kl_moment_hook": self.kl_hook} if self.kl_hook is not None else {}
        g_out = self.ns.generator_step(self.cfg, self.device, self.G, self.D, self.M, self.S, self.oG,
                                       self.oM, self.oS, self.prints, self.marks, self.ada, **kw)
        return d_out, g_out


def product_namespace(precision, ada_p=0.0):
    from types import SimpleNamespace

    import one_to_many_gan_amd as pk
    from one_to_many_gan_amd.core import training as pt
    from one_to_many_gan_amd.model import builder as pb
    from one_to_many_gan_amd.model import loss as pl

    pk.set_precision(precision)
    pt.set_async_scalars(True)  # as train.py runs the loop: the logged scalars are read at the log interval, not per step
    return SimpleNamespace(Discriminator=pb.Discriminator, Generator=pb.Generator,
                           MappingNetwork=pb.MappingNetwork, StyleExtractor=pb.StyleExtractor,
                           make_adam=pk.make_adam, ImageBuffer=pt.ImageBuffer, ADAp=pl.ADAp,
                           make_ada=(pk.IdentityADA if ada_p == 0 else
                                     (lambda: pk.AdaptiveDiscriminatorAugmentation(**pk.REFERENCE_ADA_SWITCHES))),
                           discriminator_step=pt.discriminator_step,
                           generator_step=pt.generator_step, make_throttle=pk.ops.StepThrottle)


def oracle_namespace():
    """CPU baseline leg only: the oracle is the checker / reported baseline, never shipped."""
    from types import SimpleNamespace

    from oracle import model as om
    from oracle import training as ot

    return SimpleNamespace(Discriminator=om.Discriminator, Generator=om.Generator,
                           MappingNetwork=om.MappingNetwork, StyleExtractor=om.StyleExtractor,
                           make_adam=lambda net, lr, betas: torch.optim.Adam(net.parameters(), lr=lr, betas=betas),
                           ImageBuffer=ot.ImageBuffer, ADAp=ot.ADAp, make_ada=ot.IdentityADA,
                           discriminator_step=ot.discriminator_step, generator_step=ot.generator_step,
                           make_throttle=lambda device: contextlib.nullcontext())


def kernel_profile(trainer, precision, steps=3):
    """``steps`` extra (untimed) steps with the library's launch timing on: a HIP-event pair around every
    MFMA conv KERNEL (inside the launch site, so a call that runs two kernels files them separately), on
    the stream each kernel is launched on.  Returns the dominant kernel's per-step aggregate."""
    from one_to_many_gan_amd import _hip

    from one_to_many_gan_amd import ops as o2m_ops

    def timed_steps():
        _hip.launch_timing(True)
        try:
            for _ in range(steps):
                trainer.step()
            torch.cuda.synchronize()
            table = {k: list(v) for k, v in _hip.launch_timing_read().items()}
        finally:
            _hip.launch_timing(False)
        for a in table.values():  # per step
            a[0], a[1], a[2] = a[0] / steps, a[1] / steps, a[2] / steps
        return table

    # The timed region runs THREE streams since round 3 (main, weight gradient, extraction group): a launch's
    # duration there includes whatever shares the chip with it, which says how the streams pack, not what the kernel
    # does with the chip.  The roofline figure is therefore taken in `steps` extra steps of the same binary on ONE
    # stream (every kernel alone on the chip: this is what a rocprofv3 table of a single-stream run shows too,
    # profiles/README.md); the durations of the same kernels as the timed region runs them are reported beside it
    # (`in_timed_region`).
    multi = (getattr(o2m_ops, "_WGRAD_STREAM", False) or getattr(o2m_ops, "_GROUP_STREAM", False)
             or getattr(o2m_ops, "_D_OVERLAP", False))
    in_step = timed_steps() if multi else None
    saved = (o2m_ops._WGRAD_STREAM, o2m_ops._GROUP_STREAM, o2m_ops._D_OVERLAP)
    o2m_ops._WGRAD_STREAM = o2m_ops._GROUP_STREAM = o2m_ops._D_OVERLAP = False
    try:
        agg = timed_steps()
    finally:
        o2m_ops._WGRAD_STREAM, o2m_ops._GROUP_STREAM, o2m_ops._D_OVERLAP = saved
    if not agg:
        return None
    total_conv_s = sum(a[1] for a in agg.values())
    name, (n, secs, flops) = max(agg.items(), key=lambda kv: kv[1][1])
    achieved = flops / secs / 1e12
    peak = MFMA_PEAK[precision] / 1e12
    # HBM/fabric bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE
    # with the gfx950 correction), committed under profiles/: PMC cannot be read from inside the run
    traffic = traffic_of = None
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_igemm*.json")), reverse=True):
        try:
            rec = json.load(open(path))
            if rec.get("kernel") == name:
                traffic = rec["fabric_bytes_per_launch"]
                traffic_of = {"shape": rec.get("shape"), "algorithmic_bytes": rec.get("algorithmic_bytes_per_launch"),
                              "file": os.path.basename(path)}
                break
        except (OSError, ValueError, KeyError):
            pass

    def table(t):
        return {k: {"launches": round(v[0], 1), "ms": round(v[1] * 1e3, 3), "tflops": round(v[2] / v[1] / 1e12, 1)}
                for k, v in sorted(t.items())}

    mfma_ms = sum(v[1] for k, v in agg.items() if v[2] > 0) * 1e3
    mfma_flops = sum(v[2] for v in agg.values())
    return {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_measured_on": traffic_of,
        "kernel": name, "launches_per_step": round(n, 1), "avg_launch_us": round(secs / n * 1e6, 2),
        "measured": "single-stream steps (the kernel alone on the chip)",
        "in_timed_region": (None if not in_step or name not in in_step else {
            "achieved": round(in_step[name][2] / in_step[name][1] / 1e12, 2),
            "frac": round(in_step[name][2] / in_step[name][1] / 1e12 / peak, 4),
            "avg_launch_us": round(in_step[name][1] / in_step[name][0] * 1e6, 2),
            "note": "the same kernel as the timed region runs it: beside the weight-gradient stream and the "
                    "extraction-group stream (durations include what shares the chip)"}),
        "share_of_conv_time": round(secs / total_conv_s, 3),
        "all_mfma_kernels_ms_per_step": round(mfma_ms, 3),
        "all_mfma_kernels_tflops": round(mfma_flops / (mfma_ms * 1e-3) / 1e12, 1),
        "all_conv_kernels": table(agg),
        "all_conv_kernels_in_timed_region": table(in_step) if in_step else None,
    }


def count_dispatches(trainer):
    """Operator dispatches of ONE step (a proxy for its kernel launches: every o2m:: op is one launch, two for the
    weight gradient and the InstanceNorm backward; view / metadata aten ops are not counted)."""
    from torch.utils._python_dispatch import TorchDispatchMode

    skip = ("aten.empty", "aten.as_strided", "aten.slice", "aten.detach", "aten.view", "aten.select", "aten.permute",
            "aten.t.", "aten.transpose", "aten.alias", "aten.expand", "aten.unsqueeze", "aten.record_stream",
            "aten._unsafe_view", "aten.reshape", "aten.narrow", "aten.squeeze", "aten.unbind", "aten.split", "aten.chunk",
            "aten.is_pinned", "aten._local_scalar_dense", "aten.lift_fresh", "aten.sym_")
    n = {"o2m": 0, "aten": 0}

    class Count(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if name.startswith("o2m."):
                n["o2m"] += 1
            elif not name.startswith(skip):
                n["aten"] += 1
            return func(*args, **(kwargs or {}))

    with Count():
        trainer.step()
    torch.cuda.synchronize()
    return {"o2m_ops": n["o2m"], "aten_ops": n["aten"], "note": "operator dispatches of one D+G step"}


def settle(tr, device, warmup, rounds=4):
    """Warm-up of a side leg: blocks of ``warmup`` un-synchronised steps (the host runs ahead exactly as in the timed
    steps) until a block needs no new device allocation, at most ``rounds`` blocks: with the host several steps ahead
    the caching allocator keeps growing its pool for a while, and a hipMalloc inside the timed steps can cost anything
    (249 / 495 ms per step were read this way on a busy host).  Returns the number of warm-up steps run."""
    n = 0
    for _ in range(rounds):
        before = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
        for _ in range(warmup):
            tr.step()
        n += warmup
        torch.cuda.synchronize()
        if torch.cuda.memory_stats(device).get("num_device_alloc", 0) == before:
            break
    return n


def extra_leg(args, device, precision, size, batch, steps=6, warmup=3):
    """One more configuration timed in the same run (BASELINE configs #4 and #5), a few steps each."""
    cfg = make_config(size, args.channels, batch)
    tr = Trainer(product_namespace(precision), cfg, device)
    warmup = settle(tr, device, max(warmup, steps))  # (blocks as long as the timed region: the same host lead, the same pool)
    mallocs = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"workload": f"{size}x{size}x{args.channels}, batch {batch}, {precision}", "ms_per_step": round(dt * 1e3, 3),
           "value": round(batch / dt, 3), "unit": "images/sec", "steps": steps, "warmup": warmup,
           # (hipMalloc calls inside the timed steps: a leg that is still growing its memory pool reads slow)
           "device_allocs_in_timed_steps": torch.cuda.memory_stats(device).get("num_device_alloc", 0) - mallocs,
           "reserved_gib": round(torch.cuda.max_memory_reserved(device) / 2**30, 1),
           "alloc_retries": torch.cuda.memory_stats(device).get("num_alloc_retries", 0)}
    flop = FLOP_PER_IMAGE_STEP.get((size, args.channels))
    if flop:
        out["step_mfma_frac"] = round(batch / dt * flop / MFMA_PEAK[precision], 4)
    del tr
    torch.cuda.empty_cache()
    return out


def cpu_baseline(size, channels, timed_steps=2, batch=8):
    """The CPU oracle (a restatement of the reference's own CPU path, pinned against it) timed the
    way SURVEY.md section 8d prescribes -- all host cores of the GPU's share, 1 warm-up step, then
    timed D+G steps -- on a BOUNDED sample: batch 8 instead of 16, 2 timed steps (a B = 16 step
    costs ~95 s on 16 cores, so warm-up + 3 steps would take > 6 minutes; at B = 8 the leg is ~2.5 minutes and the
    default bench.py run still finishes in about four)."""
    cores = min(len(os.sched_getaffinity(0)), 16)  # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(cores)
    tr = Trainer(oracle_namespace(), make_config(size, channels, batch), torch.device("cpu"))
    t0 = time.perf_counter()
    tr.step()  # warm-up: first-touch allocation, oneDNN primitive creation
    warm = time.perf_counter() - t0
    times = []
    for _ in range(timed_steps):
        t0 = time.perf_counter()
        tr.step()
        times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return {"value": round(batch / dt, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 warm-up + {timed_steps} timed D+G steps at batch {batch} (bounded sample of the batch-16 "
                      f"workload: half its batch; a batch-16 step is ~95 s of CPU), {size}x{size}x{channels}, fp32, CPU oracle",
            "seconds_per_step": round(dt, 2), "warmup_seconds": round(warm, 2),
            "step_seconds": [round(t, 2) for t in times]}


def graph_mode_leg(device, size=64, channels=1, batch=4, steps=20):
    """The host-bound shape (BASELINE config #1's: 64x64x1, batch 4) eagerly and as ONE replayed HIP graph
    (one_to_many_gan_amd/core/graphed.py: device-resident draws, history pool, controller and scalar sums).  Not the
    headline workload: at 256x256 the replayed three-stream schedule is slower than the eager one."""
    from one_to_many_gan_amd.core.graphed import GraphedStep

    out = {"workload": f"{size}x{size}x{channels}, batch {batch}, bf16", "steps": steps}
    for name, capture in (("eager_ms_per_step", False), ("graph_ms_per_step", True)):
        cfg = make_config(size, channels, batch)
        cfg["training"]["image_buffer_size"] = 4 * batch
        tr = Trainer(product_namespace("bf16"), cfg, device)
        gs = GraphedStep(cfg, device, {"D": tr.D, "G": tr.G, "M": tr.M, "S": tr.S},
                         {"D": tr.oD, "G": tr.oG, "M": tr.oM, "S": tr.oS}, tr.prints, tr.marks, tr.ada, capture=capture)
        for _ in range(10):  # pool fill, warm-up, capture
            gs.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gs.step()
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) / steps * 1e3, 3)
        if capture:
            out["captured"] = gs.graph is not None
        d, g = gs.logged_means()
        out.setdefault("finite", True)
        out["finite"] = out["finite"] and all(v == v for v in d + g)
        del tr, gs
        torch.cuda.empty_cache()
    return out


def parity_mode_leg(args, device, steps=6, warmup=3):
    """The SAME workload in the precision that meets the 1e-3 parity gate (fp32 storage, bf16x3 split
    MFMA), timed in this run so that the gate-passing throughput is driver-measured too."""
    cfg = make_config(args.size, args.channels, args.batch)
    tr = Trainer(product_namespace("fp32"), cfg, device)
    warmup = settle(tr, device, max(warmup, steps))  # (blocks as long as the timed region: the same host lead, the same pool)
    mallocs = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    mallocs = torch.cuda.memory_stats(device).get("num_device_alloc", 0) - mallocs
    ips = args.batch / dt
    flop = FLOP_PER_IMAGE_STEP.get((args.size, args.channels))
    out = {"dtype": "fp32 storage, bf16x3 split MFMA (outputs within 1e-3 of the CPU reference: "
                    "tests/test_hip_parity.py)", "ms_per_step": round(dt * 1e3, 3), "value": round(ips, 3),
           "unit": "images/sec", "steps": steps, "warmup": warmup, "device_allocs_in_timed_steps": mallocs,
           "reserved_gib": round(torch.cuda.max_memory_reserved(device) / 2**30, 1),
           "alloc_retries": torch.cuda.memory_stats(device).get("num_alloc_retries", 0)}
    if flop:
        out["step_mfma_frac"] = round(ips * flop / 2.5e15, 4)
    return out


def spawn_ranks(n):
    """``python bench.py --gpus N`` without a launcher: start N ranks ourselves, BEFORE this process
    touches the GPU, and relay rank 0's JSON line.  Never falls through to a single rank."""
    import socket
    import subprocess

    have = torch.cuda.device_count()  # does not initialise the GPU on this image
    share = os.environ.get("O2M_SHARE_GPU") == "1"
    if have < n and not share:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible; refusing to report a {n}-GPU number "
                         f"from fewer devices")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (weak scaling)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="bf16 = BASELINE config #2 (headline); fp32 = the 1e-3 parity mode; fp8 = config #5")
    ap.add_argument("--ada-p", type=float, default=0.0,
                    help="hold the augmentation at this probability instead of 0 (extra measurement: the "
                         "headline workload and the CPU baseline are defined at p = 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the fp32-split (1e-3 parity gate) leg")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the fp8 (config #5) and 512x512 (config #4) legs")
    ap.add_argument("--dump-params", default=None, help="write a checksum of every rank's weights (tests)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)  # does not return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # O2M_DIST_BACKEND=gloo + O2M_SHARE_GPU=1: rehearsal of the multi-rank path on ONE GPU
    # (tests/test_dist_gpu.py); the real runs use RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("O2M_DIST_BACKEND", "nccl")
    share = os.environ.get("O2M_SHARE_GPU") == "1"
    device = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    cfg = make_config(args.size, args.channels, args.batch)
    trainer = Trainer(product_namespace(args.precision, args.ada_p), cfg, device, seed_offset=rank)
    trainer.fixed_ada_p = args.ada_p
    if world > 1:
        from one_to_many_gan_amd import dist as o2m_dist

        opts = [trainer.oD, trainer.oG, trainer.oM, trainer.oS]
        o2m_dist.broadcast_parameters(opts)
        reducers = [o2m_dist.BucketReducer(o) for o in opts]
        trainer.kl_hook = o2m_dist.make_kl_moment_hook()
        o2m_dist.sync_ada_p(trainer.ada_p)
    else:
        reducers = []

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step()
    fence()
    mallocs0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = trainer.step()
    fence()
    elapsed = time.perf_counter() - t0
    mallocs = torch.cuda.memory_stats(device).get("num_device_alloc", 0) - mallocs0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    images_per_sec = args.batch * world * args.steps / elapsed
    flop = FLOP_PER_IMAGE_STEP.get((args.size, args.channels))
    out = {
        "metric": "images/sec (whole node), 256x256 G+D train step",
        "value": round(images_per_sec, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "device_allocs_in_timed_steps": mallocs,  # (hipMalloc calls of the caching allocator inside the timed steps)
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"{args.size}x{args.size}x{args.channels} D+G step (discriminator_step + "
                               f"generator_step), batch {args.batch}/GPU, stock config.toml hyper-parameters, "
                               f"ADA at p={args.ada_p:g}", "global_batch": args.batch * world,
                   "parallelism": f"dp{world}", "last_losses": {"d": float(last[0][0]), "g": float(last[1][0])}},
    }
    if flop:
        out["step_mfma_frac"] = round(images_per_sec * flop / (world * MFMA_PEAK[args.precision]), 4)
        out["algorithmic_tflops_per_gpu"] = round(images_per_sec * flop / world / 1e12, 2)
    if args.dump_params:  # (before rank 0 takes its extra, unsynchronised timer steps)
        sums = [float(o.bucket.flat.double().sum()) for o in (trainer.oD, trainer.oG, trainer.oM, trainer.oS)]
        with open(f"{args.dump_params}.rank{rank}", "w") as f:
            json.dump(sums, f)
    if rank == 0 and not args.no_kernel_profile:
        for r in reducers:  # the extra profiled steps run on rank 0 alone: no collectives of any kind
            r.enabled = False
        trainer.kl_hook = None
        if world > 1:
            trainer.ada_p.update_p = trainer.ada_p.local_update_p  # undo dist.sync_ada_p's all-reduce
        out["roofline"] = kernel_profile(trainer, args.precision)
        out["dispatches_per_step"] = count_dispatches(trainer)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_parity_mode and args.precision == "bf16":
        del trainer
        torch.cuda.empty_cache()
        out["parity_mode"] = parity_mode_leg(args, device)
        if (args.size, args.batch) == (256, 16) and not args.no_extra_legs:
            out["fp8_mode"] = extra_leg(args, device, "fp8", 256, 16)     # BASELINE config #5
            out["config4"] = extra_leg(args, device, "bf16", 512, 8)      # BASELINE config #4 (one GPU's share)
            try:   # config #1's shape on the GPU: host-bound eagerly, replayed as one HIP graph (a side leg: never fatal)
                out["graph_mode"] = graph_mode_leg(device)
            except Exception as e:  # noqa: BLE001
                out["graph_mode"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size, args.channels)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
