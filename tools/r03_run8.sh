#!/bin/bash
out=gpurun_out/r03h; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -3 $out/gputest.log
tools/ab_bench.sh -n 3 "O2M_HIP_LIB=build/variants/p8plain.so" "O2M_WGRAD_P8=0" > $out/ab.log 2>&1; cat $out/ab.log
