"""Diagnostic: is the float64 CPU oracle itself sane on this host?  (disc64 shared-mask mystery)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.cases import run_case, fp64_mode
from tests.namespaces import oracle_ns
rel = lambda a, b: float((a.double().flatten() - b.double().flatten()).norm() / (b.double().norm() + 1e-300))
for name in sys.argv[1:] or ["disc64", "disc32"]:
    gold = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    for threads in (os.cpu_count(), 1):
        torch.set_num_threads(threads)
        a = run_case(name, oracle_ns(), "cpu")
        with fp64_mode():
            b = run_case(name, oracle_ns(), "cpu")
        wa = max((rel(a[k], torch.from_numpy(gold[k])), k) for k in gold.files if not k.endswith("sum"))
        wb = max((rel(b[k], torch.from_numpy(gold[k])), k) for k in gold.files if not k.endswith("sum"))
        print(f"{name} threads={threads}: fp32 oracle vs fixture {wa[0]:.2e} ({wa[1]}) | fp64 oracle vs fixture {wb[0]:.2e} ({wb[1]})", flush=True)
