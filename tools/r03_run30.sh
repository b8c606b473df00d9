#!/bin/bash
out=gpurun_out/r03aa; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -2 $out/gputest.log
python bench.py --no-cpu-baseline --no-parity-mode --no-extra-legs --no-kernel-profile --steps 40 2>/dev/null | cut -c1-200
python tools/host_time.py 2>/dev/null; SIZE=64 BATCH=4 python tools/host_time.py 2>/dev/null
