#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-kernel-profile --no-extra-legs 2>/dev/null | python -c 'import sys, json, torch; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "allocs in timed steps", d["device_allocs_in_timed_steps"])'; }
for i in 1 2; do
echo -n "in flight 3: "; run O2M_MAX_STEPS_IN_FLIGHT=3
echo -n "in flight 2: "; run O2M_MAX_STEPS_IN_FLIGHT=2
echo -n "unbounded:   "; run O2M_MAX_STEPS_IN_FLIGHT=0
done
