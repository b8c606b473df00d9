"""CPU tests of the host-side logic of the product package (no GPU, no HIP calls)."""

import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import model as om
from oracle import training as ot

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    """The C-ABI library must load on a CPU-only box and export exactly the entry points
    include/o2m_hip.h declares (no compute calls here)."""
    from one_to_many_gan_amd import _hip

    header = open(os.path.join(ROOT, "include", "o2m_hip.h")).read()
    declared = set(re.findall(r"\b(o2m_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    lib = _hip.lib()
    ops = _hip.ops()  # the TORCH_LIBRARY shim over the same ABI: one o2m:: op per launcher
    for name in set(_hip.SIGNATURES) - _hip.MEASUREMENT_ONLY:  # (the launch timer is host-side bookkeeping, not an operator)
        assert hasattr(ops, name[len("o2m_"):]), name
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.o2m_abi_version() == _hip.ABI_VERSION
    assert lib.o2m_reduce_blocks(8 * 2048 * 3 + 8) == 4
    assert lib.o2m_instnorm_ws_floats(2, 100, 16) > 0


def test_struct_layouts_match_header():
    from one_to_many_gan_amd import _hip

    assert ctypes.sizeof(_hip.ConvDesc) == 7 * 8 + 16 * 4 + 4 * 8  # 7 pointers, 13 ints + stats_mode, fold_pad, reserved1; stats, deq_scale, aux, aux_scaled
    assert ctypes.sizeof(_hip.WgradDesc) == 5 * 8 + 14 * 4 + 8  # 5 pointers, 14 ints, slabs
    assert ctypes.sizeof(_hip.PrepJob) == 6 * 8 + 8 * 4  # o2m_prep_job: 6 pointers, 5 ints, c, first_block, reserved


def test_hot_path_fails_loudly_without_gpu():
    import one_to_many_gan_amd as pk
    from one_to_many_gan_amd.model import layers

    pk.set_precision("fp32")
    conv = layers.EqualisedConv2d(8, 8, 3, padding=1)
    with pytest.raises(RuntimeError, match="GPU"):
        conv(torch.zeros(1, 8, 5, 5))


@pytest.mark.parametrize("n", [4, 7, 8, 15, 31, 62, 63, 126, 255])
def test_banded_operators_match_torch(n):
    from one_to_many_gan_amd import resample as R

    eye = torch.eye(n, dtype=torch.float64).view(1, n, n, 1).float()  # each column = basis vector
    for kind, fn in (("blur", om.f_blur), ("up", om.f_upsample), ("down", om.f_downsample)):
        if kind == "down" and n < 2:
            continue
        # apply the oracle op along H to the identity: column j of the result = A[:, j]
        x = torch.eye(n).view(1, 1, n, n).repeat(1, 1, 1, 1)
        # use a separable probe: f(e_i e_j^T) summed over j gives A e_i * (A 1)^T; instead build A from 1-D probes
        cols = []
        for j in range(n):
            probe = torch.zeros(1, 1, n, 3)
            probe[0, 0, j, :] = 1.0  # constant along W: W-operator rows sum to 1
            out = fn(probe)
            cols.append(out[0, 0, :, 0])
        a_ref = torch.stack(cols, 1).double().numpy()
        a = R.operator_matrix(kind, n)
        assert a.shape == a_ref.shape
        assert np.abs(a - a_ref).max() < 2e-6, kind
        start, w, T = R.banded(a)
        dense = np.zeros_like(a)
        for r in range(a.shape[0]):
            dense[r, start[r]: start[r] + T] = w[r]
        assert np.abs(dense - a).max() < 1e-7
        assert (start >= 0).all() and (start + T <= a.shape[1]).all()
        st, wt, Tt = R.banded(np.ascontiguousarray(a.T))
        dense_t = np.zeros_like(a.T)
        for r in range(a.shape[1]):
            dense_t[r, st[r]: st[r] + Tt] = wt[r]
        assert np.abs(dense_t - a.T).max() < 1e-7


def test_adap_and_imagebuffer_match_oracle():
    import random

    from one_to_many_gan_amd.core.training import ImageBuffer
    from one_to_many_gan_amd.model.loss import ADAp

    a, b = ADAp(256, 5.12e-4, 4, 0.6), ot.ADAp(256, 5.12e-4, 4, 0.6)
    g = torch.Generator().manual_seed(3)
    for i in range(400):
        v = torch.rand((), generator=g) * 2 - (1.0 if i < 200 else 0.2)
        a.update_p(v)
        b.update_p(v)
        assert a() == b()
    assert a() > 0  # the schedule moved
    with pytest.raises(ValueError):
        ImageBuffer(0)
    for cls in (ImageBuffer, ot.ImageBuffer):
        random.seed(5)
        buf = cls(3)
        outs = [buf(torch.arange(4.0).view(4, 1, 1, 1) + 4 * s)[:, 0, 0, 0] for s in range(6)]
        if cls is ImageBuffer:
            mine = torch.stack(outs)
        else:
            assert torch.equal(mine, torch.stack(outs))


def test_mapping_network_rng_order_matches_oracle():
    from one_to_many_gan_amd.model.builder import MappingNetwork
    from oracle.detweights import fill_state_dict

    a, b = MappingNetwork(6, 2, 0.9), om.MappingNetwork(6, 2, 0.9)
    fill_state_dict(a, "m")
    fill_state_dict(b, "m")
    for net in (a, b):
        torch.manual_seed(11)
        outs = [net.get_single_w(3, 6, torch.device("cpu"), 1) for _ in range(6)]
        d = (torch.tensor([0.2, 0.5, 0.9]), torch.tensor([0.1, 0.4, 1.0]))
        outs += list(net.get_two_w(3, 6, torch.device("cpu"), d))
        outs.append(net.get_single_w(3, 6, torch.device("cpu"), 0))
        if net is a:
            mine = outs
        else:
            for x, y in zip(mine, outs):
                assert torch.allclose(x, y, atol=1e-6)
    assert torch.all(a(torch.randn(4, 6)) >= 0)  # final ReLU: styles are non-negative


def test_state_dict_keys_match_oracle_layout():
    """Same keys/shapes as the reference layout so checkpoints load."""
    from one_to_many_gan_amd.model import builder as pb

    pairs = [
        (pb.Generator(3, 6, (64, 64), 16, 5, 8), om.Generator(3, 6, (64, 64), 16, 5, 8)),
        (pb.Discriminator(3), om.Discriminator(3)),
        (pb.StyleExtractor(3, 6), om.StyleExtractor(3, 6)),
        (pb.MappingNetwork(6, 2, 0.9), om.MappingNetwork(6, 2, 0.9)),
    ]
    for mine, ref in pairs:
        a, b = mine.state_dict(), ref.state_dict()
        assert list(a) == list(b)
        for k in a:
            assert a[k].shape == b[k].shape, k
        mine.load_state_dict(b)
    g = pairs[0][0]
    assert g.n_style_blocks == pairs[0][1].n_style_blocks


def test_losses_small_torch_parts_match_oracle():
    from one_to_many_gan_amd.model.loss import style_cycle_loss_func

    a, b = torch.randn(5, 6), torch.randn(5, 6)
    assert torch.allclose(style_cycle_loss_func(a, b), ot.style_cycle_loss_func(a, b), atol=1e-6)
    assert torch.allclose(style_cycle_loss_func(a, b, normalise=False, cos_l2_ratio=0.5),
                          ot.style_cycle_loss_func(a, b, normalise=False, cos_l2_ratio=0.5), atol=1e-6)


def _toml_text(cfg):
    def val(v):
        if isinstance(v, bool):
            return "true" if v else "false"
        if isinstance(v, str):
            return f'"{v}"'
        if isinstance(v, (list, tuple)):
            return "[" + ", ".join(val(x) for x in v) + "]"
        return repr(v)

    out = []
    for section, kv in cfg.items():
        out.append(f"[{section}]")
        out += [f"{k} = {val(v)}" for k, v in kv.items()]
        out.append("")
    return "\n".join(out)


def test_load_config_reads_the_reference_schema(tmp_path):
    """A TOML file with the reference's sections and keys (src/data/config.py:8-68) loads into
    the nested dict the step functions take; the three directory fields become Paths."""
    from pathlib import Path

    from one_to_many_gan_amd.data.config import load_config
    from tests.cases import make_config

    cfg = make_config(1, (512, 256), 4)
    cfg["training"].update(checkpoint_directory="checkpoints", training_run="run_name")
    cfg["evaluation"] = {"log_interval": 500, "checkpoint_interval": 5000, "n_evaluation_images": 10000,
                         "inference_batch_size": 32}
    cfg["data"].update(shoemark_data_dir="/data/Shoemarks", shoeprint_data_dir="/data/Shoeprints")
    f = tmp_path / "config.toml"
    f.write_text(_toml_text(cfg))
    got = load_config(f)
    assert got["training"]["batch_size"] == 4 and got["architecture"]["w_dim"] == 6
    assert got["optimisation"]["adam_betas"] == [0.5, 0.99]
    assert got["data"]["image_size"] == [512, 256] and got["data"]["image_channels"] == 1
    assert isinstance(got["training"]["checkpoint_directory"], Path)
    assert isinstance(got["data"]["shoeprint_data_dir"], Path)
    (tmp_path / "bad.toml").write_text("[training]\nbatch_size = 4\n")
    with pytest.raises(KeyError):
        load_config(tmp_path / "bad.toml")


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` without a launcher starts N ranks itself -- and on a box with fewer
    GPUs exits non-zero instead of reporting a 1-rank number as the N-GPU one (round-1 defect)."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "O2M_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing to report a 64-GPU number" in r.stderr and "{" not in r.stdout


def test_launch_timer_switch_is_host_side_state():
    """o2m_launch_timing / o2m_launch_timing_read (bench.py's roofline source): the switch returns the previous
    state and an empty record list reads back as no kernels -- no GPU involved."""
    from one_to_many_gan_amd import _hip

    assert _hip.launch_timing(True) is False
    assert _hip.launch_timing(False) is True
    assert _hip.launch_timing_read() == {}


def test_device_resident_controller_and_history_pool_follow_the_reference_rules():
    """core/graphed.py: DeviceADAp reproduces ADAp (loss.py:11-52, window quirk included) on a scripted score trace as
    device arithmetic; DeviceImageBuffer keeps the reference's per-image rule (training.py:39-65): stored and returned
    while the pool fills, afterwards every returned image is the fresh one or one that was in the pool, the pool
    holds exactly what was put in, and about half of the images are swapped."""
    from one_to_many_gan_amd.core.graphed import DeviceADAp, DeviceImageBuffer
    from one_to_many_gan_amd.model.loss import ADAp

    g = torch.Generator().manual_seed(4)
    ref, dev = ADAp(20, 0.002, 4, 0.6), DeviceADAp(20, 0.002, 4, 0.6, torch.device("cpu"))
    peak = 0.0
    for i in range(200):
        score = torch.rand((), generator=g) * 2 - (0.2 if (i // 40) % 2 else 1.0)
        ref.update_p(score)
        dev.update_p(score)
        assert abs(ref() - dev.value()) < 1e-6, i
        peak = max(peak, ref())
    assert dev() == 0.0 and peak > 0  # (the trace moves p up and down; the device controller never makes the host wait)

    torch.manual_seed(9)
    buf = DeviceImageBuffer(6)
    seen, swapped, total = set(), 0, 0
    for step in range(40):
        batch = torch.arange(4, dtype=torch.float32).view(4, 1, 1, 1) + 4 * step + torch.zeros(4, 1, 2, 2)
        out = buf(batch)
        assert out.shape == batch.shape
        ids_in, ids_out = batch[:, 0, 0, 0].tolist(), out[:, 0, 0, 0].tolist()
        if step == 0:
            assert ids_out == ids_in and buf.num_imgs == 4
        for a, b in zip(ids_in, ids_out):
            assert b == a or b in seen, (step, a, b)   # fresh, or something stored earlier (earlier in this batch too)
            seen.add(a)
            if buf.full and step > 1:
                total += 1
                swapped += b != a
        assert set(buf.pool[: buf.num_imgs, 0, 0, 0].tolist()) <= seen
    assert 0.3 < swapped / total < 0.7
    with pytest.raises(ValueError):
        DeviceImageBuffer(0)


def test_round4_scheduling_helpers_are_inert_without_a_gpu():
    """ops.host_to_device / StepThrottle / d_step_stream (round 4): on CPU tensors they do nothing but pass the values
    through -- the gloo rehearsals and the CPU-side checks run the same step functions."""
    import torch

    from one_to_many_gan_amd import ops

    t = torch.rand(5)
    assert torch.equal(ops.host_to_device(t, "cpu"), t)
    th = ops.StepThrottle("cpu")
    for _ in range(4):
        with th:
            pass
    assert th.queue == []
    assert ops.d_step_stream(torch.device("cpu")) is None
    ops.d_step_mark(torch.device("cpu"), "done")  # no-ops
    ops.d_step_wait(torch.device("cpu"), "done")
