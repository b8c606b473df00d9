#!/bin/bash
out=gpurun_out/r03n; mkdir -p $out
for v in "" build/variants/nostore.so build/variants/staged.so; do
  echo "== ${v:-default}"; O2M_HIP_LIB=$v timeout -k 10 200 python tools/bench_conv.py 2>/dev/null | grep "64x64 256->256" || exit 1
done
