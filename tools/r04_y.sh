#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c 'import sys, json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["parity_mode"]["ms_per_step"])'; }
echo -n "default: "; run O2M_AB=1
echo -n "sync H2D: "; run O2M_ASYNC_H2D=0
echo -n "per-layer finalize: "; run O2M_BATCHED_FINALIZE=0
echo -n "no kernel profile: "; env O2M_AB=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --no-kernel-profile 2>/dev/null | python -c 'import sys, json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["parity_mode"]["ms_per_step"])'
