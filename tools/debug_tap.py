"""Debug: does recording activation masks (ops.ACT_TAP, host syncs in the forward) change the
gradients of a net-level case?  Usage: python tools/debug_tap.py [case]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests.cases import run_case
from tests.namespaces import product_ns
import one_to_many_gan_amd.ops as pops

name = sys.argv[1] if len(sys.argv) > 1 else "disc64"
gold = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
rel = lambda a, b: float((a.double().flatten() - b.double().flatten()).norm() / (b.double().norm() + 1e-300))
runs = {}
for label, tap, sync in (("plain1", False, False), ("tap", True, False), ("plain2", False, False), ("sync", False, True)):
    pops.ACT_TAP = [] if tap else None
    if sync:
        orig = pops._tap_activation
        pops._tap_activation = lambda y, act, residual: torch.cuda.synchronize()
    try:
        runs[label] = run_case(name, product_ns("fp32"), "cuda")
    finally:
        pops.ACT_TAP = None
        if sync:
            pops._tap_activation = orig
    worst = max((rel(runs[label][k], torch.from_numpy(gold[k])), k) for k in gold.files if not k.endswith("sum"))
    print(f"{label:7s} vs fixture: worst {worst[0]:.2e} ({worst[1]})", flush=True)
for a, b in (("plain1", "tap"), ("plain1", "plain2"), ("plain1", "sync")):
    worst = max((rel(runs[a][k], runs[b][k]), k) for k in runs[a] if not k.endswith("sum"))
    print(f"{a} vs {b}: worst {worst[0]:.2e} ({worst[1]})")
