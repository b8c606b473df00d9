#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04v; mkdir -p $out
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --precision fp8 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-kernel-profile --no-extra-legs 2>/dev/null | python -c 'import sys, json; print(json.loads(sys.stdin.read())["ms_per_step"])'; }
for i in 1 2; do
echo -n "fp8 scaled mfma: "; run O2M_AB=1
echo -n "fp8 old kernel : "; run O2M_HIP_LIB=build/variants/tapmajor.so
echo -n "fp8 scaled, single stream: "; run O2M_WGRAD_STREAM=0 O2M_GROUP_STREAM=0 O2M_SIDE_STYLE=0
done
python bench.py --precision fp8 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-extra-legs > $out/bench_fp8.json 2> $out/bench_fp8.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04v/bench_fp8.json'))
print(d['ms_per_step'])
for name,v in sorted(d['roofline'].get('all_conv_kernels',{}).items(), key=lambda kv:-kv[1].get('ms',0))[:14]:
    print(f"  {name:42s} {v}")
PY
