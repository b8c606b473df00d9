#!/bin/bash
# round-4 call e: generalised halo wgrad (4x4, clipped, in-scale), tiny-kernel fixes: kernel tests, parity, bench
out=$GRAFT_REPO_ROOT/gpurun_out/r04e; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider > $out/kernels.log 2>&1; rc=$?
tail -3 $out/kernels.log
[ $rc -eq 0 ] || { echo "kernel tests rc $rc"; grep -E "^E |FAILED" $out/kernels.log | head -40; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_kernels_gpu.py > $out/gputest.log 2>&1; rc=$?
tail -5 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -40; }
python bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04e/bench_default.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('dispatches_per_step'))
for name,v in sorted(d['roofline'].get('all_conv_kernels',{}).items(), key=lambda kv:-kv[1].get('ms',0)):
    print(f"  {name:42s} {v}")
PY
python tools/pointwise_bw.py > $out/pointwise_bw.txt 2>&1; head -40 $out/pointwise_bw.txt | cut -c1-150
