#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04w; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $out/gputest.log 2>&1; rc=$?
tail -3 $out/gputest.log
[ $rc -eq 0 ] || { grep -E "^E |FAILED" $out/gputest.log | head -30; exit 1; }
python bench.py > $out/bench_default.json 2> $out/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04w/bench_default.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['dispatches_per_step'], d['fp8_mode'], d['config4']['ms_per_step'], d['parity_mode']['ms_per_step'])
PY
