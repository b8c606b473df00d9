#!/bin/bash
# round-3 GPU call 2: new kernel tests first, then the suite, A/B of each fused form, rocprof table
out=gpurun_out/r03b; mkdir -p $out
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fold_scale_dot_residual or epilogue_emits or in_scale_equals or tile_kernel" > $out/newtests.log 2>&1 || { tail -60 $out/newtests.log; exit 1; }
tail -3 $out/newtests.log
python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -3 $out/gputest.log
tools/ab_bench.sh -n 2 "O2M_SPLIT_WIDE_RESAMPLE=1" "O2M_FUSED_DGRAD_DOT=0" "O2M_BLOCK_LINK=0" "O2M_WGRAD_XS=1" > $out/ab.log 2>&1; cat $out/ab.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity-mode > $GRAFT_REPO_ROOT/$out/prof.json 2> $GRAFT_REPO_ROOT/$out/prof.err
echo rocprof rc $?
