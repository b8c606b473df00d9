#!/bin/bash
# rocprofv3 PMC passes for one conv shape (each counter group in its own run, as the MI355X guide
# prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Usage: pmc_passes.sh <outdir> <one_conv args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/$tag -- python3 $GRAFT_REPO_ROOT/tools/one_conv.py "$@" > /dev/null 2>&1 || exit 1
done
