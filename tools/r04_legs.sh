#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c 'import sys, json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], {k: (d[k]["ms_per_step"], d[k].get("device_allocs_in_timed_steps"), d[k].get("reserved_gib"), d[k].get("alloc_retries")) for k in ("parity_mode","fp8_mode","config4")})'; done
