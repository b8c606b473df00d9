#!/bin/bash
# round-4 call d: new kernels (direct v2, halo wgrad, asm tr reads in wgrad p8): kernel tests, then parity suite, bench
out=$GRAFT_REPO_ROOT/gpurun_out/r04d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -v --tb=short -p no:cacheprovider > $out/kernels.log 2>&1; rc=$?
grep -E "FAILED|ERROR|passed|failed|Aborted|Fatal" $out/kernels.log | head -30
[ $rc -eq 0 ] || { echo "kernel tests rc $rc"; grep -E "^E " $out/kernels.log | head -40; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_kernels_gpu.py > $out/gputest.log 2>&1; rc=$?
tail -5 $out/gputest.log
python bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04d/bench_default.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'])
for name,v in sorted(d['roofline'].get('all_conv_kernels',{}).items(), key=lambda kv:-kv[1].get('ms',0)):
    print(f"  {name:42s} {v}")
PY
python tools/conv_shapes.py > $out/conv_shapes.txt 2>&1; grep -E "w\(64, [47]|w\(8, 4" $out/conv_shapes.txt | cut -c1-150
