#!/bin/bash
# Round-3 evidence, one gpurun call from the repo root:  gpurun --timeout 1150 -- 'bash tools/profile_round3.sh'
#   1. rocprofv3 --kernel-trace --stats of bench.py (fp32 headline, bf16) and of the dominant 48->16 layer (conv over cat((16, 32)))
#   2. --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ busy: never combined with trace domains; program directly after `--`) of that layer's
#      three passes in both dtypes, and of the bf16 16 -> 16 layer (marching kernel)
#   3. the plain bench lines (fp32 with its `secondary` object, bf16), the marching-kernel A/B, SyncBatchNorm's cost, model tables
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03_final
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_f32 -o bench_f32 -- python3 bench.py --steps 5 --warmup 4 --no-cpu-baseline --no-secondary > $O/bench_f32_under_rocprof.json 2> $O/bench_f32.err || exit 1
echo "bench f32 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_bf16 -o bench_bf16 -- python3 bench.py --steps 5 --warmup 2 --dtype bf16 > $O/bench_bf16_under_rocprof.json 2> $O/bench_bf16.err || exit 1
echo "bench bf16 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/conv_48_16 -o conv_48_16 -- python3 tools/conv_bench.py --cat 16 48 16 160 192 160 2 10 fwd,dgrad,wgrad > $O/conv_48_16.log 2>&1 || exit 1
echo "conv 48->16 stats done"
bash tools/pmc_conv.sh r03f_f32_48_16 48 16 160 192 160 2 f32 "fwd dgrad wgrad" 16 || exit 1
bash tools/pmc_conv.sh r03f_bf16_48_16 48 16 160 192 160 2 bf16 "fwd dgrad wgrad" 16 || exit 1
bash tools/pmc_conv.sh r03f_bf16_16_16 16 16 160 192 160 2 bf16 "fwd dgrad" || exit 1
cd $R
python3 bench.py > $O/final_bench_f32.json 2> $O/final_bench_f32_optable.txt || exit 1
python3 bench.py --dtype bf16 > $O/final_bench_bf16.json 2> $O/final_bench_bf16_optable.txt || exit 1
python3 tools/march_bench.py --mode march --dtype bf16 2>&1 | grep -v amdgpu.ids > $O/march_ab_march_bf16.txt || exit 1
python3 tools/march_bench.py --lib mri_epilepsy_diagnosis_amd/libmri3d_hip_nomarch.so --mode auto --dtype bf16 2>&1 | grep -v amdgpu.ids > $O/march_ab_tiled_bf16.txt || exit 1
python3 tools/syncbn_cost.py 2>&1 | grep -v amdgpu.ids > $O/syncbn_cost.txt || exit 1
TOP=70 python3 tools/model_bench.py cfg3ae 2>&1 | grep -v amdgpu.ids > $O/cfg3_autoencoder_optable.txt || exit 1
TOP=40 python3 tools/model_bench.py cfg3 2>&1 | grep -v amdgpu.ids >> $O/cfg3_autoencoder_optable.txt || exit 1
bash tools/pmc_conv.sh r03f_bf16_wgrad_16_16 16 16 160 192 160 2 bf16 "wgrad" || exit 1
for m in cfg4 m3d m3d_graph cfg3ae_graph cfg3_graph cfg5; do TOP=24 python3 tools/model_bench.py $m 2>&1 | grep -v amdgpu.ids >> $O/final_model_bench.txt || exit 1; done
cut -c1-300 $O/final_bench_f32.json; cut -c1-300 $O/final_bench_bf16.json; cat $O/syncbn_cost.txt
