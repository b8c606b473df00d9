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
