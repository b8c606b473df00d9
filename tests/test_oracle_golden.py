"""Pins the CPU oracle: every fixture under tests/golden/ was produced by the reference
itself (tools/make_golden.py, build container) and the oracle must reproduce it."""

import os

import numpy as np
import pytest
import torch

from tests.cases import CASES, SLOW_CASES, host_threads, run_case
from tests.namespaces import oracle_ns

# fp32 CPU vs fp32 CPU, same ATen kernels in a different call order: 1e-5 relative
RTOL = 1e-5


def _rel(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("name", [n for n in CASES])
def test_oracle_matches_reference_fixture(name, golden_dir):
    torch.set_num_threads(host_threads())
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    got = run_case(name, oracle_ns(), "cpu")
    assert set(got) == set(gold.files)
    tol = 2e-4 if name.startswith("steps") else RTOL  # Adam + 2 steps amplifies fp32 rounding
    for k in gold.files:
        g = torch.from_numpy(gold[k])
        assert got[k].shape == g.shape, k
        if g.abs().max() == 0:
            assert got[k].abs().max() < 1e-6, k
            continue
        assert _rel(got[k], g) <= tol, (k, _rel(got[k], g))
