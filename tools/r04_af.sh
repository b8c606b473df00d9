#!/bin/bash
cd $GRAFT_REPO_ROOT
S="48 64 64 256 128 3 1 0  32 64 64 256 128 3 1 0  16 64 64 256 128 3 1 0  48 64 64 256 256 3 1 0  32 64 64 256 256 3 1 0 16 64 64 256 256 3 1 0"
python tools/time_conv.py fwd $S 2>&1 | grep -v amdgpu.ids
