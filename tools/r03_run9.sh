#!/bin/bash
out=gpurun_out/r03i; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -3 $out/gputest.log
python bench.py --no-cpu-baseline --no-parity-mode > $out/bench.json 2> $out/bench.err; python -c "
import json; d=json.load(open('$out/bench.json')); print(d['ms_per_step'], d.get('dispatches_per_step'));r=d['roofline']
print({k:v for k,v in r.items() if not k.startswith('all_conv')})
for k,v in r['all_conv_kernels'].items(): print(k, v)"
tools/ab_bench.sh -n 2 "O2M_WGRAD_P8=0" > $out/ab.log 2>&1; cat $out/ab.log
