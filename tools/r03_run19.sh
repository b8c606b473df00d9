#!/bin/bash
out=$PWD/gpurun_out/r03s; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_fp32 -- python3 $root/bench.py --precision fp32 --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode --no-kernel-profile --no-extra-legs > $out/bench_fp32.json 2> $out/bench_fp32.err
cat $out/bench_fp32.json | cut -c1-300
f=$(ls $out/prof_fp32/*/*_kernel_stats.csv); head -40 $f | cut -d, -f1-5
