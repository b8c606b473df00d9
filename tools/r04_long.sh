#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04long; mkdir -p $out
cd $GRAFT_REPO_ROOT
(echo "== bf16, 500 steps (train.run: discriminator-step stream, two steps in flight, all streams)"; python tools/long_run.py 500 2>&1 | grep -v amdgpu.ids | tail -14
 echo "== fp8, 300 steps"; LONG_RUN_PRECISION=fp8 python tools/long_run.py 300 2>&1 | grep -v amdgpu.ids | tail -10) > $out/long_run.txt 2>&1
tail -30 $out/long_run.txt
