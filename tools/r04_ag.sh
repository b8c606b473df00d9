#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04ag; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_train_loop_gpu.py tests/test_graphed_gpu.py tests/test_dist_gpu.py -q --tb=short -p no:cacheprovider -x -k "steps or train or graph or dist or data_parallel or losses" > $out/gputest.log 2>&1; rc=$?
tail -3 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -30; exit 1; }
bash tools/ab_bench.sh -n 3 "O2M_SIDE_MAPPING=0" > $out/ab.log 2>&1; cat $out/ab.log
