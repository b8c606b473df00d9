#!/bin/bash
# Interleaved same-box A/B of bench.py variants (run it inside ONE gpurun call: boxes differ by ~5 %, the first
# run after idle is ~1 ms slower, run-to-run noise is ~+-0.3 ms).
#   tools/ab_bench.sh [-n PAIRS] "O2M_SIDE_STREAM=1" "O2M_HIP_LIB=build/variants/x.so" ...
# Each argument is one variant: a space-separated list of VAR=value settings ("" = the defaults).  The
# defaults are run before every variant, so every variant has its own adjacent baseline.
pairs=3
if [ "$1" = "-n" ]; then pairs=$2; shift 2; fi
run() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode --no-kernel-profile --no-extra-legs 2>/dev/null |
        python -c 'import sys, json; print(json.loads(sys.stdin.read())["ms_per_step"])'; }
echo -n "warm-up  "; run O2M_AB=warm
for i in $(seq $pairs); do
  for v in "$@"; do
    echo -n "base      "; run O2M_AB=base
    echo -n "[$v]  "; run O2M_AB=variant $v
  done
done
