#!/bin/bash
# round-4 first call: new parity tests + the whole gpu suite + default bench (baseline of this round)
out=$GRAFT_REPO_ROOT/gpurun_out/r04a; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/gputest.log 2>&1; rc=$?
tail -15 $out/gputest.log
[ $rc -eq 0 ] || exit 1
python bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
python -c "import json; d=json.load(open('$out/bench_default.json')); print(d['ms_per_step'], d['value'], d['cpu_baseline'])"
