#!/usr/bin/env python3
"""Per-launch counter averages of ONE kernel from the passes tools/pmc_passes.sh wrote.

  tools/pmc_summary.py <pmc_dir> <kernel-name substring> <timer name> <shape text> <algorithmic bytes> <flops> > out.json

Every pass directory holds rocprofv3's *_counter_collection.csv (one row per dispatch, counter and instance); values are
summed over instances per dispatch and averaged over the dispatches of the kernel (the first launch, which includes
first-touch effects, is dropped).  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts half the bytes of wide
coalesced reads (MI355X_MICROARCH.md, HBM section): fabric bytes = 2 * FETCH_SIZE + WRITE_SIZE."""
import collections
import csv
import glob
import json
import os
import sys

root, needle, timer_name, shape, alg_bytes, flops = sys.argv[1:7]
vals = collections.defaultdict(lambda: collections.defaultdict(float))  # counter -> dispatch -> sum
dur = {}
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    grp = path[len(root):].strip("/").split("/")[0]
    for r in csv.DictReader(open(path)):
        if needle not in r["Kernel_Name"]:
            continue
        key = (grp, r["Dispatch_Id"])
        vals[r["Counter_Name"]][key] += float(r["Counter_Value"])
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            dur[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {"kernel": timer_name, "shape": shape}
for name, per in sorted(vals.items()):
    ds = sorted(per, key=lambda k: (k[0], int(k[1])))
    first = {}
    keep = []
    for k in ds:  # drop the first dispatch of each pass
        if k[0] not in first:
            first[k[0]] = k
            continue
        keep.append(k)
    keep = keep or ds
    out[name] = round(sum(per[k] for k in keep) / len(keep), 1)
if dur:
    out["avg_launch_us_profiled"] = round(sum(dur.values()) / len(dur), 1)
f, w = out.get("FETCH_SIZE"), out.get("WRITE_SIZE")
if f is not None and w is not None:
    out["fabric_bytes_per_launch"] = int(f * 1024 * 2 + w * 1024)
out["algorithmic_bytes_per_launch"] = int(float(alg_bytes))
h, m = out.get("TCC_HIT_sum"), out.get("TCC_MISS_sum")
if h is not None and m is not None and h + m > 0:
    out["L2_hit_rate"] = round(h / (h + m), 3)
wc = out.get("SQ_WAVE_CYCLES")
if wc:
    out["wait_fraction"] = round(out.get("SQ_WAIT_ANY", 0) / wc, 3)
    out["issue_stall_fraction"] = round(out.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
if out.get("SQ_INSTS_MFMA"):
    out["VALU_per_MFMA"] = round(out.get("SQ_INSTS_VALU", 0) / out["SQ_INSTS_MFMA"], 2)
if out.get("avg_launch_us_profiled"):
    out["tflops_profiled"] = round(float(flops) / out["avg_launch_us_profiled"] / 1e6, 1)
out["collected_with"] = ("tools/pmc_passes.sh <out> <one_conv args>: five separate `rocprofv3 --kernel-trace --pmc <group> "
                         "--output-format csv -- python3 tools/one_conv.py ...` passes; tools/pmc_summary.py: per-launch averages "
                         "summed over all XCDs / SEs; FETCH_SIZE / WRITE_SIZE in KiB; fabric_bytes_per_launch = FETCH_SIZE * 1024 "
                         "* 2 (gfx950 correction) + WRITE_SIZE * 1024")
print(json.dumps(out, indent=1))
