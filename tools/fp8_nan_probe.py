"""Run N fp8 steps and report the first step whose logged losses are not finite (probe for the intermittent NaN)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tr = bench.Trainer(bench.product_namespace(os.environ.get("PROBE_PRECISION", "fp8")), bench.make_config(256, 3, 16), torch.device("cuda:0"))
bad = None
for i in range(n):
    d, g = tr.step()
    vals = [float(d[0]), float(g[0])] + [float(v) for v in g[1]]
    if not all(math.isfinite(v) for v in vals):
        bad = (i, vals); break
print("first non-finite step:", bad)
