"""GPU busy analysis of a rocprofv3 kernel trace (…_kernel_trace.csv): union of kernel intervals, time with >= 2 kernels
resident, idle gaps, per-stream (queue) busy time -- over the window of the last `--steps` bench steps if --window-ms is
given, else the whole trace."""
import csv, sys, collections, argparse
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--tail-ms", type=float, default=0.0)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows]
ev.sort()
t_end = max(e[1] for e in ev)
if a.tail_ms:
    ev = [e for e in ev if e[0] >= t_end - a.tail_ms * 1e6]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, _, _ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth, last, busy, multi, hist = 0, t0, 0, 0, collections.Counter()
for t, d in pts:
    if depth > 0: busy += t - last
    if depth > 1: multi += t - last
    hist[min(depth, 4)] += t - last
    depth += d; last = t
span = t1 - t0
print(f"kernels {len(ev)}  span {span/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms = {busy/span:.3f}  >=2 resident {multi/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms")
print("depth histogram (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
perq = collections.defaultdict(float)
for s, e, _, q in ev: perq[q] += e - s
print("per queue busy (ms):", {q: round(v / 1e6, 2) for q, v in perq.items()})
# idle gaps by the kernel that FOLLOWS them
gaps = collections.Counter(); depth = 0
cur_end = t0
for s, e, n, _ in ev:
    if s > cur_end: gaps[n[:60]] += s - cur_end
    cur_end = max(cur_end, e)
print("largest idle-gap followers:")
for n, g in gaps.most_common(12): print(f"  {g/1e3:9.1f} us  {n}")
