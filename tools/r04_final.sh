#!/bin/bash
# round-4 measurement set: default bench (JSON line), rocprof kernel tables of the default command and of the
# single-stream run, PMC passes of the dominant kernel and of this round's new MFMA kernels
out=$GRAFT_REPO_ROOT/gpurun_out/r04z; mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
echo "default bench done"; python -c "import json; d=json.load(open('$out/bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_default -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity-mode --no-extra-legs > $out/bench_profiled.json 2> $out/prof_default.err
echo "rocprof default rc $?"
O2M_WGRAD_STREAM=0 O2M_GROUP_STREAM=0 O2M_D_OVERLAP=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_single -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity-mode --no-kernel-profile --no-extra-legs > $out/bench_profiled_single.json 2> $out/prof_single.err
echo "rocprof single-stream rc $?"
cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh $out/pmc_p8 fwd 48 64 64 256 256 3 1 1 && python tools/pmc_summary.py $out/pmc_p8 conv_igemm_p8_kernel "conv_igemm_p8<bf16,256x256>" "3x3 conv 256->256, 64x64, B=48 (the decode group), reflect pad 1, bf16" 202506240 231928233984 > $out/pmc_igemm_p8.json; echo pmc p8 $?
bash tools/pmc_passes.sh $out/pmc_wgrad_p8 wgrad 48 64 64 256 256 3 1 1 && python tools/pmc_summary.py $out/pmc_wgrad_p8 conv_wgrad_p8_kernel "conv_wgrad_p8<bf16,256x256>" "weight gradient of the 3x3 256->256 layer, 64x64, B=48, reflect pad 1, bf16" 203685888 231928233984 > $out/pmc_wgrad_p8.json; echo pmc wgrad p8 $?
bash tools/pmc_passes.sh $out/pmc_wgrad_halo wgrad 48 256 256 128 64 3 1 0 && python tools/pmc_summary.py $out/pmc_wgrad_halo conv_wgrad_halo_kernel "conv_wgrad_halo<bf16,64x9x64>" "weight gradient of the 3x3 128->64 layer, 256x256, B=48, zero pad 1, bf16" 1208107008 463856467968 > $out/pmc_wgrad_halo.json; echo pmc wgrad halo $?
bash tools/pmc_passes.sh $out/pmc_halo4 fwd 16 127 127 64 128 4 1 0 && python tools/pmc_summary.py $out/pmc_halo4 conv_halo_w4c_kernel "conv_halo<bf16,4x4,8x32x64>" "4x4 conv 64->128, 127x127 -> 126x126, B=16 (discriminator trunk), zero pad 1, bf16" 98066432 66588770304 > $out/pmc_halo4.json; echo pmc halo4 $?
ls $out | head -30
