#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04u; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_hip_parity.py -q --tb=short -p no:cacheprovider -x -k "fp8" > $out/gputest.log 2>&1; rc=$?
tail -3 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -30; exit 1; }
S="48 64 64 256 256 3 1 1  32 64 64 256 256 3 1 1  16 64 64 256 256 3 1 1  48 64 64 256 256 3 2 0  48 128 128 128 256 3 1 0"
python tools/time_conv.py fwd $S 2>&1 | grep -v amdgpu.ids | tee $out/bf16.txt
python tools/time_conv.py fp8 $S 2>&1 | grep -v amdgpu.ids | tee $out/fp8.txt
python bench.py --precision fp8 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-extra-legs --no-kernel-profile 2>/dev/null | python -c 'import sys, json; print("fp8 step", json.loads(sys.stdin.read())["ms_per_step"])'
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-extra-legs --no-kernel-profile 2>/dev/null | python -c 'import sys, json; print("bf16 step", json.loads(sys.stdin.read())["ms_per_step"])'
