#!/bin/bash
# round 4, call p: the four-wave Co = 128 halo kernel: kernel + parity tests, same-box A/B against the 8-wave form,
# per-kernel table
out=$GRAFT_REPO_ROOT/gpurun_out/r04p; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_hip_parity.py -q --tb=short -p no:cacheprovider -x -k "full_size or halo or epilogue or steps256 or gen64 or dot or c256" > $out/gputest.log 2>&1; rc=$?
tail -4 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -30; exit 1; }
bash tools/ab_bench.sh -n 3 "O2M_HALO128_W4=0" > $out/ab.log 2>&1; cat $out/ab.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-extra-legs > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04p/bench.json'))
print(d['ms_per_step'])
for name,v in sorted(d['roofline'].get('all_conv_kernels',{}).items(), key=lambda kv:-kv[1].get('ms',0))[:10]:
    print(f"  {name:42s} {v}")
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-parity-mode --no-kernel-profile --no-extra-legs > $out/trace_bench.json 2> $out/trace_bench.err
cd $GRAFT_REPO_ROOT
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_busy.py $f --tail-ms 200 > $out/busy.txt 2>&1; cat $out/busy.txt
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id"]
keep = [k for k in keep if k in rows[0]]
t_end = max(int(r["End_Timestamp"]) for r in rows)
w = csv.writer(open("gpurun_out/r04p/trace_tail.csv", "w"))
w.writerow(keep)
for r in rows:
    if int(r["Start_Timestamp"]) >= t_end - 130e6:
        w.writerow([r[k][:90] if k == "Kernel_Name" else r[k] for k in keep])
PY
rm -rf $out/trace
