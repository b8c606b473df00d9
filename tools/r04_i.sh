#!/bin/bash
# round-4 call i: whole gpu suite (incl. the data-parallel equivalence test) + default bench with cpu baseline
out=$GRAFT_REPO_ROOT/gpurun_out/r04i; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $out/gputest.log 2>&1; rc=$?
tail -6 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -40; }
python bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
python -c "import json; d=json.load(open('$out/bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline'], d.get('dispatches_per_step')); print({k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.items() if k in ('parity_mode','fp8_mode','config4')})"
