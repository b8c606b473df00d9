#!/bin/bash
# round-4 call c: whole gpu suite (verbose: the log names the test that was running if the process dies), default bench,
# per-shape conv table, aten census
out=$GRAFT_REPO_ROOT/gpurun_out/r04c; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -v --tb=short -p no:cacheprovider > $out/gputest.log 2>&1; rc=$?
grep -E "FAILED|ERROR|passed|failed|Aborted|Fatal" $out/gputest.log | head -30
[ $rc -eq 0 ] || echo "pytest rc $rc"
python bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
python -c "import json; d=json.load(open('$out/bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('dispatches_per_step'))"
python tools/conv_shapes.py > $out/conv_shapes.txt 2>&1; head -5 $out/conv_shapes.txt
python tools/aten_census.py > $out/aten_census.txt 2>&1; tail -3 $out/aten_census.txt
