#!/usr/bin/env python3
"""Measure the REFERENCE's own bf16 error: run tests/cases.py on the reference under
torch.autocast(cpu, bfloat16) and compare with its fp32 fixtures (tests/golden/*.npz).
Writes tests/golden/bf16_yardstick.json = {case: {key: rel_l2}}.  The bf16 parity tests
accept the HIP path's bf16 mode when it is within a small factor of these numbers: they are
what bf16 storage costs the reference itself (BASELINE.md section 2 quotes the same probe).
Build container only (imports /root/reference through tools/make_golden.py)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from make_golden import reference_ns
from tests.cases import CASES, run_case

SKIP = {"adap", "imagebuffer", "mapping", "losses", "steps64", "steps256", "steps128"}  # steps: the reference itself raises under autocast (lerp dtype)
torch.set_num_threads(os.cpu_count() or 1)
ns = reference_ns()
PATH = os.path.join(ROOT, "tests", "golden", "bf16_yardstick.json")
only = sys.argv[1:]  # given cases are merged into the existing file
out = json.load(open(PATH)) if only else {}
for name in (only or CASES):
    if name in SKIP:
        continue
    gold = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    with torch.autocast("cpu", dtype=torch.bfloat16):
        got = run_case(name, ns, "cpu")
    d = {}
    for k in gold.files:
        if k.endswith("/sum") or k.endswith("/sqsum"):
            continue
        w = torch.from_numpy(gold[k]).double().flatten()
        g = got[k].double().flatten()
        d[k] = float((g - w).norm() / (w.norm() + 1e-30)) if w.norm() > 0 else 0.0
    out[name] = d
    worst = sorted(d.items(), key=lambda kv: -kv[1])[:5]
    print(f"{name:18s} max={worst[0][1]:.2e} | " + ", ".join(f"{k}={v:.2e}" for k, v in worst), flush=True)
json.dump(out, open(PATH, "w"), indent=0, sort_keys=True)
