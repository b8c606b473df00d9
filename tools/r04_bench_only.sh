#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04z; mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
python -c "
import json; d=json.load(open('$out/bench_default.json'))
print(d['ms_per_step'], d['value'], d['step_mfma_frac'], d['roofline']['frac'], d['device_allocs_in_timed_steps'])
for k in ('parity_mode','fp8_mode','config4','graph_mode','dispatches_per_step'): print(k, d[k])
print(d['cpu_baseline']['value'])"
