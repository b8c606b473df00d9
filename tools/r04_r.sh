#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04r; mkdir -p $out
cd $GRAFT_REPO_ROOT
python tools/cpu_issue.py > $out/cpu_issue.txt 2>&1; head -4 $out/cpu_issue.txt
bash tools/ab_bench.sh -n 3 "O2M_ASYNC_H2D=0" > $out/ab.log 2>&1; cat $out/ab.log
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_train_loop_gpu.py tests/test_graphed_gpu.py -q --tb=short -p no:cacheprovider -x  > $out/gputest.log 2>&1; tail -3 $out/gputest.log
