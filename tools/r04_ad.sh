#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04ad; mkdir -p $out
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh -n 3 "O2M_D_OVERLAP=0" > $out/ab.log 2>&1; cat $out/ab.log
timeout -k 10 1000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $out/gputest.log 2>&1; rc=$?
tail -3 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -30; exit 1; }
