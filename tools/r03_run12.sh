#!/bin/bash
out=gpurun_out/r03m; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_hip_parity.py -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -3 $out/gputest.log
timeout -k 10 300 python tools/bench_conv.py > $out/conv_direct.txt 2>&1 && O2M_HIP_LIB=build/variants/staged.so timeout -k 10 300 python tools/bench_conv.py > $out/conv_staged.txt 2>&1
grep -i "p8" $out/conv_direct.txt; echo ---; grep -i "p8" $out/conv_staged.txt
tools/ab_bench.sh -n 3 "O2M_HIP_LIB=build/variants/staged.so" > $out/ab.log 2>&1; cat $out/ab.log
