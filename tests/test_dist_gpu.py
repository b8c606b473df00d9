"""GPU rehearsal of the data-parallel path: two ranks share the single GPU of the test box
and exchange gradients over gloo (RCCL refuses two ranks on one device), which exercises the
real BucketReducer code path on HIP streams -- all-reduce launched from autograd hooks on
the side stream, event hand-off to the fused Adam, KL moment hook, parameter broadcast --
that the 8-GPU scaling run uses with backend "nccl"."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_two_ranks_stay_in_lockstep(tmp_path):
    env = dict(os.environ, O2M_DIST_BACKEND="gloo", O2M_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dump = str(tmp_path / "params")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "64", "--batch", "2",
           "--no-cpu-baseline", "--dump-params", dump]  # with the roofline leg: rank 0 then works alone
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["value"] > 0
    # the per-kernel timer steps run on rank 0 alone while rank 1 waits at the barrier: they must not issue a
    # collective (gradient reducers, KL hook and the ADAp score sync are all switched off for them)
    assert out["roofline"]["kernel"].startswith("conv_")
    a = json.load(open(dump + ".rank0"))
    b = json.load(open(dump + ".rank1"))
    # different data per rank, identical weights after 4 synchronised optimiser steps
    assert a == b, (a, b)


@pytest.mark.timeout(600)
def test_train_loop_under_data_parallelism(tmp_path):
    """train.run's data-parallel branch end to end -- broadcast, four reducers + KL hook + ADAp sync in one
    collective order on both ranks, rank-0-only logging and checkpoints while rank 1 moves on, resume from rank 0's
    checkpoint -- on two ranks sharing the GPU over gloo.  Identical weights on both ranks after 2 steps and after
    the resumed third; only rank 0 wrote files."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29613", os.path.join(ROOT, "tests", "dp_train_worker.py"),
           str(tmp_path)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    a = json.load(open(tmp_path / "rank0.json"))
    b = json.load(open(tmp_path / "rank1.json"))
    assert a["after2"] == b["after2"] and a["after3"] == b["after3"], (a, b)
    assert a["after2"] != a["after3"] and a["step"] == b["step"] == 3.0
    assert a["logged"] == 3 and b["logged"] == 0  # 2 + 1 log lines, rank 0 only
    # the dead biases ahead of InstanceNorm (D / S model.3/7/11.bias, G encoder.1/4/8.bias) get no
    # AccumulateGrad node; unless the conv's backward reports them complete, their bucket segment -- and, segments
    # going out strictly last to first, every earlier one -- waits for FusedAdam.step instead of overlapping
    # backward.  On the real networks every segment of D and S must have been launched from inside backward.
    for rank_out in (a, b):
        logs = rank_out["launch_logs"]
        assert logs["D"] and logs["S"] and logs["G"], logs
        assert set(logs["D"]) == {"hook"} and set(logs["S"]) == {"hook"}, logs
        assert logs["G"].count("hook") >= len(logs["G"]) - 1, logs
    models = sorted(p.name for p in (tmp_path / "dp" / "models").iterdir())
    assert models == ["1.tar", "2.tar", "3.tar"]
    assert len((tmp_path / "dp" / "log").read_text().splitlines()) == 3


@pytest.mark.timeout(600)
def test_data_parallel_step_equals_the_big_batch_step(tmp_path):
    """a D+G step on two ranks (local batch 2: early finalisation, the bucket reducers launching inside
    backward on their comm stream, the group stream, the global-batch KL hook) hands every optimiser the gradient of the
    single-process batch-4 step on the same four samples -- data parallelism is exact for this step -- to fp32 rounding
    (parity mode; every random draw tabled by global sample: tests/dp_equiv_worker.py)."""
    import torch

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "dp_equiv_worker.py")
    r = subprocess.run([sys.executable, worker, str(tmp_path), "single"], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", worker, str(tmp_path), "dp"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    one = torch.load(tmp_path / "single0.pt", weights_only=True)
    a, b = (torch.load(tmp_path / f"dp{k}.pt", weights_only=False) for k in (0, 1))
    for k in ("D", "G", "M", "S"):
        assert torch.equal(a[k], b[k]), k  # both ranks hold the same reduced gradient
        ref = one[k]
        err = float((a[k] - ref).norm() / ref.norm())
        # fp32-split MFMA products summed over a different partition of the samples (per-rank slabs, then the all-reduce);
        # ReLU masks are per sample and identical on both sides
        assert err < 2e-4, (k, err)
    assert abs(float(a["kl"]) - float(one["kl"])) <= 1e-5 * abs(float(one["kl"]))  # the KL of the GLOBAL batch on each rank
    assert set(a["launch_logs"]["D"]) == {"hook"} and set(a["launch_logs"]["S"]) == {"hook"}, a["launch_logs"]
