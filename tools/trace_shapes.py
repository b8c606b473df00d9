#!/usr/bin/env python3
"""Per-(kernel, grid) durations of a rocprofv3 kernel trace: tools/trace_shapes.py <kernel_trace.csv> <steps> [filter]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("unsigned short", "bf16").replace("at::native::", "")
    if flt and flt not in name:
        continue
    key = (name[:64], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = collections.Counter({k: sum(v) for k, v in agg.items()})
print(f"total {sum(tot.values()) / steps / 1e3:.3f} ms/step")
for k, t in tot.most_common(int(sys.argv[4]) if len(sys.argv) > 4 else 60):
    v = agg[k]
    print(f"{k[0]:64s} ({k[1]},{k[2]},{k[3]}) n/step={len(v) / steps:5.1f} avg={sum(v) / len(v):8.1f}us tot/step={t / steps / 1e3:6.3f}ms")
