#!/bin/bash
out=gpurun_out/r03k; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -3 $out/gputest.log
tools/ab_bench.sh -n 4 "O2M_SIDE_STYLE=0" > $out/ab.log 2>&1; cat $out/ab.log
