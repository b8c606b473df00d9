#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04aa; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_hip_parity.py -q --tb=short -p no:cacheprovider -x -k "full_size or halo or epilogue or steps or gen64 or dot or c256 or config4 or trunk or disc or style" > $out/gputest.log 2>&1; rc=$?
tail -4 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -30; exit 1; }
bash tools/ab_bench.sh -n 3 "O2M_HALO_W4C=0" > $out/ab.log 2>&1; cat $out/ab.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-extra-legs > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04aa/bench.json'))
print(d['ms_per_step'])
for name,v in sorted(d['roofline'].get('all_conv_kernels',{}).items(), key=lambda kv:-kv[1].get('ms',0))[:10]:
    print(f"  {name:42s} {v}")
PY
