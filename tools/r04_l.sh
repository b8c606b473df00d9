#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04l; mkdir -p $out
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh -n 3 "O2M_BORDER_DGRAD=1" "O2M_WGRAD_STREAM=0 O2M_GROUP_STREAM=0" > $out/ab.log 2>&1; cat $out/ab.log
