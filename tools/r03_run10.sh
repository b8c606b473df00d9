#!/bin/bash
out=gpurun_out/r03j; mkdir -p $out
timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fused_instance_norm or batched_weight or phase_pipelined" > $out/newtests.log 2>&1 || { tail -60 $out/newtests.log; exit 1; }
tail -3 $out/newtests.log
python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -3 $out/gputest.log
tools/ab_bench.sh -n 3 "O2M_FUSED_NORM_DOWN=0" "O2M_WGRAD_P8=1" > $out/ab.log 2>&1; cat $out/ab.log
