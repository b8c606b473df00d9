#!/bin/bash
out=gpurun_out/r03g; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export O2M_WGRAD_STREAM=0 O2M_GROUP_STREAM=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity-mode --no-kernel-profile > $GRAFT_REPO_ROOT/$out/prof.json 2> $GRAFT_REPO_ROOT/$out/prof.err
echo rocprof rc $?; cat $GRAFT_REPO_ROOT/$out/prof.json | head -c 600
