#!/bin/bash
# round-4 call g: where does the border data gradient lose?  kernel tables with and without it; then tests of the fused loss ops
out=$GRAFT_REPO_ROOT/gpurun_out/r04g; mkdir -p $out
cd $GRAFT_REPO_ROOT
for v in 1 0; do
  O2M_BORDER_DGRAD=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-extra-legs > $out/bench_border$v.json 2> $out/bench_border$v.err || { tail -20 $out/bench_border$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open('gpurun_out/r04g/bench_border$v.json'))
print("BORDER=$v", d['ms_per_step'], d.get('dispatches_per_step'))
for name,v in sorted(d['roofline'].get('all_conv_kernels',{}).items(), key=lambda kv:-kv[1].get('ms',0))[:8]:
    print(f"  {name:42s} {v}")
for name,v in d['roofline'].get('all_conv_kernels',{}).items():
    if 'border' in name: print(name, v)
PY
done
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -q --tb=short -p no:cacheprovider -k "steps or losses or disc" > $out/gputest.log 2>&1; tail -5 $out/gputest.log
