#!/bin/bash
out=gpurun_out/r03v; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "wgrad or beyond" > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -2 $out/gputest.log
for v in "" build/variants/head.so; do
  echo "== ${v:-default}"; O2M_HIP_LIB=$v timeout -k 10 200 python tools/bench_conv.py wgrad 2>/dev/null | grep "64x64 256->256 k3" || exit 1
done
tools/ab_bench.sh -n 4 "O2M_WGRAD_P8=1" "O2M_WGRAD_STREAM=0" > $out/ab.log 2>&1; cat $out/ab.log
