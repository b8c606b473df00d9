#!/bin/bash
# full GPU suite, then the round-4 measurement set (tools/r04_final.sh) in the same call
out=$GRAFT_REPO_ROOT/gpurun_out/r04z; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $out/gputest.log 2>&1; rc=$?
tail -3 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -30; exit 1; }
rm -rf $out/prof_default $out/prof_single $out/pmc_*
bash tools/r04_final.sh
