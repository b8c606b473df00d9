"""What each kernel family costs on the WALL CLOCK of the default (three-stream) step: time the step with the family's
launches skipped (results are garbage; only the timing is read).  A family whose kernel time is mostly hidden beside
other streams saves little when skipped."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from one_to_many_gan_amd import _hip as H

FAMILIES = {
    "none": [],
    "wgrad": ["conv2d_wgrad"],
    "conv_fwd_dgrad": ["conv2d_fwd"],
    "fold+act_bwd": ["fold_scale_dot", "act_bwd_reduce"],
    "instnorm": ["instnorm_apply", "instnorm_bwd", "instnorm_stats", "instnorm_act_resample2d", "instnorm_resample_bwd"],
    "resample": ["resample2d"],
    "style+reduce": ["style_bwd", "style_fwd", "reduce_fwd", "reduce_bwd"],
}
args = type("A", (), dict(size=256, channels=3, batch=16))()
dev = torch.device("cuda:0")
cfg = bench.make_config(args.size, args.channels, args.batch)
tr = bench.Trainer(bench.product_namespace("bf16"), cfg, dev)
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
orig = {n: getattr(H, n) for fam in FAMILIES.values() for n in fam}
base = None
for fam, names in list(FAMILIES.items()) + [("none", [])]:
    for n in names:
        setattr(H, n, lambda *a, **k: None)
    for _ in range(2):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    for n in names:
        setattr(H, n, orig[n])
    if fam == "none" and base is None:
        base = ms
    print(f"skip {fam:16s}: {ms:7.2f} ms/step   (saves {base - ms:6.2f})", flush=True)
