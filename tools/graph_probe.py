"""Feasibility probe: capture one D+G step in a HIP graph (torch.cuda.graph) and time its replay.  NOT a product path:
the CPU-side random draws are replaced by device draws and the history pool is bypassed, so that the step has no host
dependence -- what a graph-capturable step function would have to provide."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from one_to_many_gan_amd.core import training as pt

size, batch = int(os.environ.get("SIZE", 256)), int(os.environ.get("BATCH", 16))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg = bench.make_config(size, 3 if size > 64 else 1, batch)
tr = bench.Trainer(bench.product_namespace("bf16"), cfg, dev)
pt.set_async_scalars(False)

# host dependence out: device RNG for every draw, no style mixing decision on the host, no history pool, static batch
fixed_p, fixed_m = next(tr.prints), next(tr.marks)   # (the pools are built on the first draw, with the CPU generator)
_randn, _rand = torch.randn, torch.rand
torch.randn = lambda *a, **k: _randn(*a, **{**k, "device": dev})
torch.rand = lambda *a, **k: _rand(*a, **k) if "generator" in k or a == ((),) else _rand(*a, **{**k, "device": dev})
tr.M.style_mixing_prob = 0.0
tr.buffer = lambda images: images.detach()
tr.prints = iter(lambda: fixed_p, None)
tr.marks = iter(lambda: fixed_m, None)
pt._floats = lambda *scalars: [0.0] * len(scalars)   # (the scalar read-back would be a memcpy node into a static buffer)
tr.ada_p.update_p = lambda s: None
tr.ada_p.__class__.__call__ = lambda self: 0.0

def timed(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for _ in range(4):
    tr.step()
print(f"eager: {timed(tr.step):.2f} ms/step", flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        tr.step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        tr.step()
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:600]); sys.exit(0)
print("captured", flush=True)
print(f"replay: {timed(g.replay):.2f} ms/step")
