#!/bin/bash
# round-3 GPU call 1: full -m gpu suite, A/B of the batched decoder passes, default bench, rocprof kernel table
out=gpurun_out/r03a; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1 || { tail -40 $out/gputest.log; exit 1; }
tail -3 $out/gputest.log
tools/ab_bench.sh -n 2 "O2M_BATCH_DECODES=0" > $out/ab.log 2>&1; cat $out/ab.log
python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity-mode > $GRAFT_REPO_ROOT/$out/prof.json 2> $GRAFT_REPO_ROOT/$out/prof.err
echo rocprof rc $?
