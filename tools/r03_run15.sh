#!/bin/bash
out=gpurun_out/r03p; mkdir -p $out
for v in "" build/variants/tapouter.so build/variants/tapouter_recompute.so; do
  echo "== ${v:-default}"; O2M_HIP_LIB=$v timeout -k 10 200 python tools/bench_conv.py 2>/dev/null | grep "64x64 256->256" || exit 1
done
root=$PWD
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $root/$out/pmc/$tag -- python3 $root/tools/one_conv.py fwd 48 64 64 256 256 3 1 1 > /dev/null 2>&1 || echo "pmc $tag failed"
done
cd $root; python tools/pmc_summary.py $out/pmc conv_igemm_p8_kernel "conv_igemm_p8<bf16,256x256>" "chunk-outer" 202506240 231928233984 > $out/pmc_chunk_outer.json; cat $out/pmc_chunk_outer.json
