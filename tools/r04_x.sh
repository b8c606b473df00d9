#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --precision fp32 --steps 6 --warmup 3 --no-cpu-baseline --no-parity-mode --no-kernel-profile --no-extra-legs 2>/dev/null | python -c 'import sys, json; print(json.loads(sys.stdin.read())["ms_per_step"])'; }
echo -n "fp32 default: "; run O2M_AB=1
echo -n "fp32 sync H2D: "; run O2M_ASYNC_H2D=0
echo -n "fp32 per-layer finalize: "; run O2M_BATCHED_FINALIZE=0
echo -n "fp32 default again: "; run O2M_AB=1
