#!/bin/bash
# round-4 call f: border dgrad + style_bwd fix: kernel tests, parity, then same-box A/B of the new switches
out=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider > $out/kernels.log 2>&1; rc=$?
tail -3 $out/kernels.log
[ $rc -eq 0 ] || { echo "kernel tests rc $rc"; grep -E "^E |FAILED" $out/kernels.log | head -40; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_kernels_gpu.py > $out/gputest.log 2>&1; rc=$?
tail -5 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -40; exit 1; }
bash tools/ab_bench.sh -n 3 "O2M_BORDER_DGRAD=0" "O2M_WGRAD_HALO=0" "O2M_CONV_DIRECT=0" > $out/ab.log 2>&1; cat $out/ab.log
