#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh -n 3 "O2M_MAIN_PRIO=1" "O2M_MAIN_PRIO=1 O2M_PRIO_GROUP=-1" "O2M_PRIO_GROUP=-1"
