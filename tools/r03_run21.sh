#!/bin/bash
out=gpurun_out/r03t; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dist_gpu.py -m gpu -x -q -k "async or deterministic or dist or train" > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -2 $out/gputest.log
tools/ab_bench.sh -n 4 "O2M_ASYNC_SCALARS=0" > $out/ab.log 2>&1; cat $out/ab.log
python tools/host_time.py 2>/dev/null; SIZE=32 BATCH=2 python tools/host_time.py 2>/dev/null
