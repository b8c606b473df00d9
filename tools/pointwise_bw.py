"""Achieved bytes/s of every non-MFMA wrapper call of one single-stream D+G step: bytes = the tensor arguments of the call
(each read or written once by these kernels), time = a HIP-event pair around the call."""
import os, sys, collections
os.environ.setdefault("O2M_WGRAD_STREAM", "0"); os.environ.setdefault("O2M_GROUP_STREAM", "0"); os.environ.setdefault("O2M_SIDE_STYLE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from one_to_many_gan_amd import _hip as H

NAMES = ["fold_scale_dot", "act_bwd_reduce", "instnorm_apply", "instnorm_bwd", "instnorm_stats", "instnorm_act_resample2d",
         "instnorm_resample_bwd", "resample2d", "reduce_fwd", "reduce_bwd", "pack_nchw", "modulate_weights", "style_bwd",
         "style_fwd", "wgrad_finalize", "instnorm_finalize", "conv2d_dots_finalize", "prepare_weights_batched"]
args = type("A", (), dict(size=256, channels=3, batch=16))()
dev = torch.device("cuda:0")
tr = bench.Trainer(bench.product_namespace("bf16"), bench.make_config(256, 3, 16), dev)
for _ in range(4):
    tr.step()
torch.cuda.synchronize()
log = []
def wrap(name, fn):
    def w(*a, **k):
        ts = [t for t in list(a) + list(k.values()) if isinstance(t, torch.Tensor)]
        nbytes = sum(t.numel() * t.element_size() for t in ts)
        big = max(ts, key=lambda t: t.numel()) if ts else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        log.append((name, nbytes, tuple(big.shape) if big is not None else (), e0, e1))
        return r
    return w
for n in NAMES:
    setattr(H, n, wrap(n, getattr(H, n)))
tr.step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for name, nb, shape, e0, e1 in log:
    k = (name, shape)
    agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1) * 1e-3; agg[k][2] += nb
tot = sum(v[1] for v in agg.values())
print(f"total {tot*1e3:.2f} ms in {len(log)} calls")
for (name, shape), (n, t, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{name:26s} {str(shape):28s} n={n:3d} {t*1e3:7.3f} ms  {nb/n/1e6:8.1f} MB/call  {nb/t/1e12:5.2f} TB/s")
