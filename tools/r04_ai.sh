#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh -n 3 "O2M_SIDE_MAPPING=0" "O2M_PRIO_MAP=0"
