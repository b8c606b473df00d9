#!/bin/bash
# kernel trace of the default step -> busy / idle analysis + the tail of the trace for offline study
out=$GRAFT_REPO_ROOT/gpurun_out/r04s; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-parity-mode --no-kernel-profile --no-extra-legs > $out/trace_bench.json 2> $out/trace_bench.err
cd $GRAFT_REPO_ROOT
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_busy.py $f --tail-ms 200 > $out/busy.txt 2>&1; cat $out/busy.txt
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [k for k in ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id"] if k in rows[0]]
t_end = max(int(r["End_Timestamp"]) for r in rows)
w = csv.writer(open("gpurun_out/r04s/trace_tail.csv", "w"))
w.writerow(keep)
for r in rows:
    if int(r["Start_Timestamp"]) >= t_end - 130e6:
        w.writerow([r[k][:90] if k == "Kernel_Name" else r[k] for k in keep])
PY
rm -rf $out/trace
python -c "import json; print(json.load(open('gpurun_out/r04s/trace_bench.json'))['ms_per_step'])"
