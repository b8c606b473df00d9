#!/bin/bash
# round 4, call o: same-box A/B (batched finalize off; Co = 128 halo layers as two 64-channel blocks) + a kernel trace
# of the default three-stream step for the busy analysis (tools/trace_busy.py)
out=$GRAFT_REPO_ROOT/gpurun_out/r04o; mkdir -p $out
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh -n 3 "O2M_BATCHED_FINALIZE=0" "O2M_HALO128_SPLIT=1" > $out/ab.log 2>&1; cat $out/ab.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-parity-mode --no-kernel-profile --no-extra-legs > $out/trace_bench.json 2> $out/trace_bench.err
cd $GRAFT_REPO_ROOT
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_busy.py $f --tail-ms 200 > $out/busy.txt 2>&1; cat $out/busy.txt
rm -rf $out/trace
