#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r04q; mkdir -p $out
cd $GRAFT_REPO_ROOT
python tools/cpu_issue.py > $out/cpu_issue.txt 2>&1; head -5 $out/cpu_issue.txt
