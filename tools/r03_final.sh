#!/bin/bash
# round-3 measurement set: default bench (JSON line), rocprof kernel tables of the default command and of the
# single-stream run, PMC passes of the dominant kernel and the two new MFMA kernels
out=$GRAFT_REPO_ROOT/gpurun_out/r03zz; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1 || { tail -60 $out/gputest.log; exit 1; }
tail -2 $out/gputest.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
echo "default bench done"; python -c "import json; d=json.load(open('$out/bench_default.json')); print(d['ms_per_step'], d['value'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_default -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity-mode > $out/bench_profiled.json 2> $out/prof_default.err
echo "rocprof default rc $?"
O2M_WGRAD_STREAM=0 O2M_GROUP_STREAM=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_single -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity-mode --no-kernel-profile > $out/bench_profiled_single.json 2> $out/prof_single.err
echo "rocprof single-stream rc $?"
cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh $out/pmc_p8 fwd 48 64 64 256 256 3 1 1 && python tools/pmc_summary.py $out/pmc_p8 conv_igemm_p8_kernel "conv_igemm_p8<bf16,256x256>" "3x3 conv 256->256, 64x64, B=48 (the decode group), reflect pad 1, bf16" 202506240 231928233984 > $out/pmc_igemm_p8.json; echo pmc p8 $?
bash tools/pmc_passes.sh $out/pmc_halo64 fwd 48 256 256 128 64 3 1 0 && python tools/pmc_summary.py $out/pmc_halo64 conv3x3_halo_kernel "conv3x3_halo<bf16,8x32x64>" "3x3 conv 128->64, 256x256, B=48, zero pad 1, bf16" 1208107008 463856467968 > $out/pmc_halo64.json; echo pmc halo $?
bash tools/pmc_passes.sh $out/pmc_wgrad_p8 wgrad 48 64 64 256 256 3 1 1 && python tools/pmc_summary.py $out/pmc_wgrad_p8 conv_wgrad_p8_kernel "conv_wgrad_p8<bf16,256x256>" "weight gradient of the 3x3 256->256 layer, 64x64, B=48, reflect pad 1, bf16" 203685888 231928233984 > $out/pmc_wgrad_p8.json; echo pmc wgrad $?
ls -la $out | head -30
