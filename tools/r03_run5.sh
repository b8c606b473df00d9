#!/bin/bash
out=gpurun_out/r03e; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "full_size_bf16 or epilogue_emits or halo_kernel" > $out/newtests.log 2>&1 || { tail -60 $out/newtests.log; exit 1; }
tail -3 $out/newtests.log
python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -3 $out/gputest.log
tools/ab_bench.sh -n 3 "O2M_CONV_HALO=0" > $out/ab.log 2>&1; cat $out/ab.log
python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/bench.err; python -c "
import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step']); 
for k,v in d['roofline']['all_conv_kernels'].items(): print(k, v)"
