#!/bin/bash
# round-4 call k: clipped halo forward kernel (4x4 trunk): kernel tests, parity, A/B
out=$GRAFT_REPO_ROOT/gpurun_out/r04k; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider > $out/kernels.log 2>&1; rc=$?
tail -3 $out/kernels.log
[ $rc -eq 0 ] || { echo "kernel tests rc $rc"; grep -E "^E |FAILED" $out/kernels.log | head -40; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_kernels_gpu.py > $out/gputest.log 2>&1; rc=$?
tail -4 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -40; exit 1; }
bash tools/ab_bench.sh -n 3 "O2M_CONV_HALO4=0" > $out/ab.log 2>&1; cat $out/ab.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-extra-legs > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04k/bench.json'))
print(d['ms_per_step'])
for name,v in sorted(d['roofline'].get('all_conv_kernels',{}).items(), key=lambda kv:-kv[1].get('ms',0)):
    print(f"  {name:42s} {v}")
PY
