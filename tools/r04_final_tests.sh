#!/bin/bash
# full GPU suite, then the default bench line of the final build
out=$GRAFT_REPO_ROOT/gpurun_out/r04z; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $out/gputest.log 2>&1; rc=$?
tail -3 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -30; exit 1; }
bash tools/r04_bench_only.sh
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
