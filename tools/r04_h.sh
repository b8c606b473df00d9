#!/bin/bash
# round-4 call h: border kernel v3 (K-split over waves), LDS tap tables, fused loss ops: tests, kernel table, A/B
out=$GRAFT_REPO_ROOT/gpurun_out/r04h; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider > $out/kernels.log 2>&1; rc=$?
tail -3 $out/kernels.log
[ $rc -eq 0 ] || { echo "kernel tests rc $rc"; grep -E "^E |FAILED" $out/kernels.log | head -40; exit 1; }
O2M_BORDER_DGRAD=1 timeout -k 10 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_kernels_gpu.py > $out/gputest_border.log 2>&1; rc=$?
tail -4 $out/gputest_border.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest_border.log | head -40; exit 1; }
O2M_BORDER_DGRAD=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-extra-legs > $out/bench_border1.json 2> $out/bench_border1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04h/bench_border1.json'))
print("BORDER=1", d['ms_per_step'], d.get('dispatches_per_step'))
for name,v in d['roofline'].get('all_conv_kernels',{}).items():
    if 'border' in name or '128x128' in name or 'p8' in name: print(name, v)
PY
bash tools/ab_bench.sh -n 3 "O2M_BORDER_DGRAD=1" > $out/ab.log 2>&1; cat $out/ab.log
python tools/pointwise_bw.py > $out/pointwise_bw.txt 2>&1; grep -E "instnorm_resample_bwd|style_bwd|total" $out/pointwise_bw.txt | cut -c1-150
