"""Debug probe: does the phase-pipelined weight gradient compute, IN the training step (own stream, beside everything
else), what it computes alone from the same operands?  Operands are snapshotted on the PRODUCER stream at the point the
launch is ordered behind (clones there; nothing is added in front of the kernel on the weight-gradient stream); the in-flow
contribution of a call is the sum of the slab partials it wrote (read back on its stream after the kernels)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from one_to_many_gan_amd import _hip as H
prec = os.environ.get("PROBE_PRECISION", "bf16")
tr = bench.Trainer(bench.product_namespace(prec), bench.make_config(256, 3, 16), torch.device("cuda:0"))
from one_to_many_gan_amd import ops
rec = []
orig = H.conv2d_wgrad
wst = None
def wg(x, gy, dw, **k):
    big = x.shape[1:] == (64, 64, 256) and gy.shape[-1] == 256 and k.get("in_scale") is None
    if big:
        cur = torch.cuda.current_stream()           # = the weight-gradient stream here
        prod = torch.cuda.default_stream()          # producers run on the main stream in this configuration
        with torch.cuda.stream(prod):
            xc, gc = x.clone(), gy.clone()          # snapshots taken on the producer stream, at ITS current position
    r = orig(x, gy, dw, **k)
    if big:
        ws = H._SLABS.get(x.device, 1)
        n = dw.numel()
        rec.append((xc, gc, ws[: 28 * n].view(28, -1).sum(0).view(dw.shape).clone(), dict(k)))
    return r
H.conv2d_wgrad = wg
bad_total = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    rec.clear()
    tr.step()
    torch.cuda.synchronize()
    for n, (xc, gc, inflow, k) in enumerate(rec):
        ref = torch.zeros_like(inflow)
        orig(xc, gc, ref, **k)
        torch.cuda.synchronize()
        if not bool(torch.isfinite(inflow).all()) or float((ref - inflow).abs().max()) > 1e-5 * max(1.0, float(ref.abs().max())):
            d = (ref - inflow)
            fin = bool(torch.isfinite(inflow).all())
            print(f"step {i} call {n} {tuple(xc.shape)}: in-flow != alone; in-flow finite {fin}; max|diff| {float(d.abs().max()) if fin else float('nan')}; max|ref| {float(ref.abs().max())}; "
                  f"elements off by > 1e-5: {int(((ref - inflow).abs() > 1e-5).sum())}, non-finite {int((~torch.isfinite(inflow)).sum())}; where (co, tap, ci of the first): {[int(v) for v in torch.nonzero((~torch.isfinite(inflow)) | ((ref - inflow).abs() > 1e-5))[0].tolist()] if (bool((~torch.isfinite(inflow)).any()) or bool(((ref - inflow).abs() > 1e-5).any())) else None}")
            bad_total += 1
print(prec, "calls whose in-flow result differs from the result computed alone:", bad_total)
