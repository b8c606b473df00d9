#!/bin/bash
out=gpurun_out/r03c; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -3 $out/gputest.log
tools/ab_bench.sh -n 2 "O2M_SPLIT_WIDE_RESAMPLE=1" "O2M_FUSED_DGRAD_DOT=0" "O2M_BLOCK_LINK=0" "O2M_WGRAD_INSCALE=1" > $out/ab.log 2>&1; cat $out/ab.log
python tools/aten_census.py > $out/census.txt 2>&1; tail -75 $out/census.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity-mode > $GRAFT_REPO_ROOT/$out/prof.json 2> $GRAFT_REPO_ROOT/$out/prof.err
echo rocprof rc $?
