#!/bin/bash
out=gpurun_out/r03y; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_hip_parity.py -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -2 $out/gputest.log
tools/ab_bench.sh -n 4 "O2M_NORM_DOWN_TILE=0" > $out/ab.log 2>&1; cat $out/ab.log
timeout -k 10 300 python tools/pointwise_bw.py 2>/dev/null | grep "instnorm_act_resample2d\|total"
