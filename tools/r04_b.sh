#!/bin/bash
# round-4 call b: isolate the config4 abort, see the failures of the routed tests, first run of the direct kernels
out=$GRAFT_REPO_ROOT/gpurun_out/r04b; mkdir -p $out
cd $GRAFT_REPO_ROOT
echo "== config4 alone, direct kernels off"; O2M_CONV_DIRECT=0 timeout -k 10 300 python -m pytest tests/test_hip_parity.py -k config4 -q --tb=short > $out/cfg4_nodirect.log 2>&1; echo rc $?; tail -3 $out/cfg4_nodirect.log
echo "== direct kernel tests"; timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -k "direct_conv or p8_kernel_instance" -v --tb=short > $out/direct.log 2>&1; echo rc $?; grep -E "PASS|FAIL|Error|assert" $out/direct.log | head -40
echo "== config4 alone, direct kernels on"; timeout -k 10 300 python -m pytest tests/test_hip_parity.py -k config4 -q --tb=short > $out/cfg4_direct.log 2>&1; echo rc $?; tail -3 $out/cfg4_direct.log
echo "== routed / full-size block tests"; timeout -k 10 600 python -m pytest tests/test_hip_parity.py -k "routed or full_size_blocks or c256_b16" -v --tb=short > $out/routed.log 2>&1; echo rc $?; grep -E "PASS|FAIL|Error|assert|^E " $out/routed.log | head -60
