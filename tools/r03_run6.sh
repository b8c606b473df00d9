#!/bin/bash
out=gpurun_out/r03f; mkdir -p $out
timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "phase_pipelined_weight or wgrad" > $out/newtests.log 2>&1 || { tail -60 $out/newtests.log; exit 1; }
tail -3 $out/newtests.log
python tools/bench_conv.py all > $out/bench_conv.txt 2>&1; cat $out/bench_conv.txt
O2M_CONV_HALO=0 O2M_WGRAD_P8=0 python tools/bench_conv.py all > $out/bench_conv_r2kernels.txt 2>&1; cat $out/bench_conv_r2kernels.txt
python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -3 $out/gputest.log
tools/ab_bench.sh -n 3 "O2M_WGRAD_P8=0" > $out/ab.log 2>&1; cat $out/ab.log
