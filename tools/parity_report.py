"""Print per-tensor relative errors of the HIP path against the reference fixtures.
Usage: python tools/parity_report.py [--prec fp32,bf16] [--shared-masks] [case ...]
--shared-masks: fp32 mode vs the fp64 oracle replaying the HIP path's activation masks
(tests/cases.py::run_case_shared_masks), with the count of mask elements that differed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests.cases import CASES, run_case
from tests.namespaces import product_ns

args = [a for a in sys.argv[1:] if not a.startswith("--")]
precs = ["fp32", "bf16"]  # also "fp8" (config #5) via --prec=fp8
for a in sys.argv[1:]:
    if a.startswith("--prec="):
        precs = a.split("=")[1].split(",")
names = args or [n for n in CASES if not n.startswith("steps")]
if "--shared-masks" in sys.argv:
    from tests.cases import run_case_shared_masks
    from tests.namespaces import oracle_ns

    for name in (args or [n for n in CASES if n.startswith(("gen", "disc", "style"))]):
        got, want, flipped, total = run_case_shared_masks(name, product_ns("fp32"), oracle_ns())
        errs = sorted(((float((got[k].double().flatten() - w.double().flatten()).norm() / (w.double().norm() + 1e-300)), k)
                       for k, w in want.items() if not k.endswith(("/sum", "/sqsum")) and float(w.abs().max()) > 0),
                      reverse=True)
        gerr = [e for e in errs if e[1].startswith("g")]
        print(f"{name:14s} masks flipped {flipped}/{total} ({flipped / max(total, 1):.1e}) | worst grad "
              f"{gerr[0][0]:.2e} ({gerr[0][1]}) | worst output {max((e for e in errs if not e[1].startswith('g')), default=(0, ''))[0]:.2e} | "
              + ", ".join(f"{k}={e:.1e}" for e, k in errs[:5]), flush=True)
    sys.exit(0)
for name in names:
    gold = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    for prec in precs:
        t0 = time.time()
        try:
            got = run_case(name, product_ns(prec), "cuda")
        except Exception as e:  # noqa: BLE001
            print(f"{name} [{prec}] EXCEPTION {type(e).__name__}: {e}")
            continue
        worst = []
        for k in gold.files:
            w = torch.from_numpy(gold[k]).double().flatten()
            g = got[k].double().flatten()
            if k.endswith("/sum") or k.endswith("/sqsum"):
                continue
            err = float((g - w).norm() / (w.norm() + 1e-30))
            worst.append((err, k))
        worst.sort(reverse=True)
        top = ", ".join(f"{k}={e:.2e}" for e, k in worst[:6])
        print(f"{name:18s} [{prec}] {time.time()-t0:5.1f}s max={worst[0][0]:.2e} | {top}", flush=True)
