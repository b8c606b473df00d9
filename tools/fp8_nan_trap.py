"""Debug probe: after every phase-pipelined weight-gradient call (on its stream, AFTER the kernels): are the slab
partials it wrote finite, is the accumulator finite?"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from one_to_many_gan_amd import _hip as H
tr = bench.Trainer(bench.product_namespace("fp8"), bench.make_config(256, 3, 16), torch.device("cuda:0"))
log = []
orig = H.conv2d_wgrad
def wg(x, gy, dw, **k):
    r = orig(x, gy, dw, **k)
    if x.shape[1:] == (64, 64, 256) and gy.shape[-1] == 256:
        ws = H._SLABS.get(x.device, 1)
        n = 28 * 256 * 2304
        sl = ws[:n].view(28, -1)
        log.append((len(log), tuple(x.shape), torch.isfinite(dw).all(), torch.isfinite(sl).all(dim=1), sl.abs().amax(dim=1), dw.abs().max()))
    return r
H.conv2d_wgrad = wg
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    log.clear()
    tr.step()
    okg = torch.isfinite(tr.oG.bucket.grad).all()
    torch.cuda.synchronize()
    if not bool(okg):
        print("step", i, "G gradient bucket not finite;", len(log), "p8 wgrad calls")
        for n, shp, okdw, oksl, mxsl, mxdw in log:
            if not bool(okdw) or not bool(oksl.all()):
                badslices = [j for j in range(28) if not bool(oksl[j])]
                print("  call", n, shp, "dw finite", bool(okdw), "max|dw|", float(mxdw), "non-finite slices", badslices, "max|slab| of the finite ones", float(mxsl[oksl].max()) if bool(oksl.any()) else None)
        break
else:
    print("no non-finite gradient")
