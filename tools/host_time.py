"""Host enqueue time of one D+G step (no blocking read inside the measured loop) beside its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = type("A", (), dict(size=int(os.environ.get("SIZE", 256)), channels=3, batch=int(os.environ.get("BATCH", 16))))()
dev = torch.device("cuda:0")
cfg = bench.make_config(args.size, args.channels, args.batch)
tr = bench.Trainer(bench.product_namespace("bf16"), cfg, dev)
for _ in range(5):
    tr.step()
torch.cuda.synchronize()
tr.ada_p.__class__.__call__ = lambda self: 0.0   # the one blocking read of the loop body (train.py:206)
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); e0.record()
for _ in range(n):
    tr.step()
t1 = time.perf_counter(); e1.record()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"size {args.size} batch {args.batch}: host enqueue {1e3*(t1-t0)/n:.2f} ms/step, GPU {e0.elapsed_time(e1)/n:.2f} ms/step, wall {1e3*(t2-t0)/n:.2f} ms/step")
