#!/bin/bash
# round 4, call n: full GPU suite on the batched-finalize build, then same-box A/B of (batched finalize off) and
# (Co = 128 halo layers as two 64-channel blocks)
out=$GRAFT_REPO_ROOT/gpurun_out/r04n; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $out/gputest.log 2>&1; rc=$?
tail -4 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -30; exit 1; }
O2M_HALO128_SPLIT=1 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_hip_parity.py -q --tb=short -p no:cacheprovider -k "full_size or halo or epilogue or steps256 or gen64 or dot" > $out/gputest_split.log 2>&1; rc=$?
tail -2 $out/gputest_split.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest_split.log | head -30; exit 1; }
bash tools/ab_bench.sh -n 3 "O2M_BATCHED_FINALIZE=0" "O2M_HALO128_SPLIT=1" > $out/ab.log 2>&1; cat $out/ab.log
