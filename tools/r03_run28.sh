#!/bin/bash
out=gpurun_out/r03x; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_hip_parity.py tests/test_torch_ops.py -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -2 $out/gputest.log
tools/ab_bench.sh -n 4 "O2M_PATH_TAP=0" > $out/ab.log 2>&1; cat $out/ab.log
