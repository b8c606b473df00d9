"""Per-shape table of the conv2d_fwd calls (forward and data-gradient GEMMs) of one single-stream D+G step: count, time (HIP-event pair
around the wrapper call) and algorithmic TFLOP/s by (x shape, filter shape, pad, stride, fold_pad, dot epilogue)."""
import os, sys, collections
os.environ.setdefault("O2M_WGRAD_STREAM", "0"); os.environ.setdefault("O2M_GROUP_STREAM", "0"); os.environ.setdefault("O2M_SIDE_STYLE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from one_to_many_gan_amd import _hip as H
tr = bench.Trainer(bench.product_namespace("bf16"), bench.make_config(256, 3, 16), torch.device("cuda:0"))
for _ in range(3): tr.step()
torch.cuda.synchronize()
log = []
orig = H.conv2d_fwd
def w(x, wt, y, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(x, wt, y, **k); e1.record()
    log.append((tuple(x.shape), tuple(wt.shape), k.get("pad"), k.get("stride", 1), k.get("fold_pad", 0), k.get("aux") is not None, e0, e1))
    return r
H.conv2d_fwd = w
tr.step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for xs, ws, pad, st, fp, dot, e0, e1 in log:
    agg[(xs, ws, pad, st, fp, dot)][0] += 1; agg[(xs, ws, pad, st, fp, dot)][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"{len(log)} conv2d_fwd calls, {tot:.2f} ms")
for (xs, ws, pad, st, fp, dot), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    B, Hh, Ww, Ci = xs; Co = ws[-4]; kh, kw = ws[-3], ws[-2]
    S = max(st, 1); ho, wo = (Hh + 2 * pad - kh) // S + 1, (Ww + 2 * pad - kw) // S + 1
    fl = 2.0 * B * ho * wo * Co * kh * kw * Ci * n
    print(f"x{xs} w{ws} pad{pad} s{st} fold{fp} dot{int(dot)} n={n} {t:7.3f} ms {fl/t/1e9:7.1f} TF/s")
