#!/bin/bash
# round-4 call j: fp32 parity mode on the non-spilling tiles: parity tests in fp32 + A/B of the mode's step time
out=$GRAFT_REPO_ROOT/gpurun_out/r04j; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -k "fp32 or f32 or shared or adjoint" > $out/gputest.log 2>&1; rc=$?
tail -4 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -40; exit 1; }
for i in 1 2; do
for v in 0 1; do
  echo -n "O2M_F32_BIG_TILES=$v  "; O2M_F32_BIG_TILES=$v python bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode --no-kernel-profile --no-extra-legs 2>/dev/null | python -c 'import sys, json; print(json.loads(sys.stdin.read())["ms_per_step"])'
done; done
