#!/bin/bash
cd $GRAFT_REPO_ROOT
S="32 256 256 128 64 3 1 0  48 256 256 128 64 3 1 0  16 256 256 128 64 3 1 0  16 127 127 64 128 4 1 0  16 126 126 128 64 4 2 0  16 30 30 512 256 4 2 0 16 63 63 128 256 4 1 0"
echo "== four-wave"; python tools/time_conv.py fwd $S 2>&1 | grep -v amdgpu.ids
echo "== eight-wave"; O2M_HALO_W4C=0 python tools/time_conv.py fwd $S 2>&1 | grep -v amdgpu.ids
bash tools/ab_bench.sh -n 3 "O2M_HALO_W4C=1" 
