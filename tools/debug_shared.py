"""Debug the disc64 shared-mask mismatch: which side (HIP path with tap / fp64 oracle with replay) leaves the fixture?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.cases import run_case_shared_masks
from tests.namespaces import oracle_ns, product_ns
rel = lambda a, b: float((a.double().flatten() - b.double().flatten()).norm() / (b.double().norm() + 1e-300))
for name in sys.argv[1:]:
    gold = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    got, want, fl, tot = run_case_shared_masks(name, product_ns("fp32"), oracle_ns())
    ks = [k for k in gold.files if not k.endswith("sum")]
    wg = max((rel(got[k], torch.from_numpy(gold[k])), k) for k in ks)
    ww = max((rel(want[k], torch.from_numpy(gold[k])), k) for k in ks)
    gw = max((rel(got[k], want[k]), k) for k in ks)
    print(f"{name}: flips {fl}/{tot} | HIP vs fixture {wg[0]:.2e} ({wg[1]}) | replayed fp64 oracle vs fixture {ww[0]:.2e} ({ww[1]}) | HIP vs oracle {gw[0]:.2e}", flush=True)
