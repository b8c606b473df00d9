#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04t; mkdir -p $out
cd $GRAFT_REPO_ROOT
S="48 64 64 256 256 3 1 1  32 64 64 256 256 3 1 1  16 64 64 256 256 3 1 1  48 64 64 256 256 3 2 0  32 64 64 256 256 3 2 0  48 128 128 128 256 3 1 0  32 128 128 256 128 3 1 0"
echo "== chunk-major (default build)"; python tools/time_conv.py fwd $S 2>&1 | grep -v amdgpu.ids | tee $out/chunk_major.txt
echo "== tap-major (variant)"; O2M_HIP_LIB=build/variants/tapmajor.so python tools/time_conv.py fwd $S 2>&1 | grep -v amdgpu.ids | tee $out/tap_major.txt
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_hip_parity.py -q --tb=short -p no:cacheprovider -x -k "full_size or p8 or epilogue or steps256 or gen64 or dot or c256 or fp8" > $out/gputest.log 2>&1; rc=$?
tail -3 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -30; exit 1; }
bash tools/ab_bench.sh -n 3 "O2M_HIP_LIB=build/variants/tapmajor.so" > $out/ab.log 2>&1; cat $out/ab.log
