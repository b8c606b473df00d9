"""Host-side cost of queuing one D+G step: wall time of tr.step() calls with NO synchronisation between them (the time
the host needs to issue a step; when it is close to the device's step time the device starves in the phases made of
small kernels), then a cProfile of a few steps (tottime ranking)."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
tr = bench.Trainer(bench.product_namespace("bf16"), bench.make_config(256, 3, 16), torch.device("cuda:0"))
for _ in range(4): tr.step()
torch.cuda.synchronize()
ts = []
t00 = time.perf_counter()
for _ in range(12):
    t0 = time.perf_counter(); tr.step(); ts.append((time.perf_counter() - t0) * 1e3)
t_issue = (time.perf_counter() - t00) * 1e3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t00) * 1e3
print("issue ms per step:", [round(t, 1) for t in ts])
print(f"12 steps: issued in {t_issue:.1f} ms, finished in {t_all:.1f} ms ({t_all / 12:.2f} ms/step)")
# host-only: how long does issuing take when the device is never the limit?  (queue depth is finite, so a host that runs
# ahead blocks in hipLaunchKernel: the first steps after a sync are the unblocked ones)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(4): tr.step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40); print(s.getvalue()[:8000])
