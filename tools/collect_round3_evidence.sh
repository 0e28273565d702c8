#!/bin/bash
# Copy what tools/profile_round3.sh left under gpurun_out/ into profiles/ (run here, after the gpurun call came back).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof_r03_final
P=profiles
cp $O/final_bench_f32.json $P/r03_final_bench_f32.json
cp $O/final_bench_bf16.json $P/r03_final_bench_bf16.json
grep -v amdgpu.ids $O/final_bench_f32_optable.txt > $P/r03_final_bench_f32_optable.txt
grep -v amdgpu.ids $O/final_bench_bf16_optable.txt > $P/r03_final_bench_bf16_optable.txt
cp $O/bench_f32_under_rocprof.json $P/r03_final_bench_f32_under_rocprof.json
cp $O/bench_bf16_under_rocprof.json $P/r03_final_bench_bf16_under_rocprof.json
cp $O/final_model_bench.txt $P/r03_final_model_bench.txt
cp $O/bench_f32/bench_f32_kernel_stats.csv $P/r03_final_rocprofv3_kernel_stats_bench_f32_steps5.csv
cp $O/bench_bf16/bench_bf16_kernel_stats.csv $P/r03_final_rocprofv3_kernel_stats_bench_bf16_steps5.csv
cp $O/conv_48_16/conv_48_16_kernel_stats.csv $P/r03_final_rocprofv3_kernel_stats_conv_48_16_layer.csv
cp $O/march_ab_march_bf16.txt $P/r03_march_ab_march_bf16.txt
cp $O/march_ab_tiled_bf16.txt $P/r03_march_ab_tiled_bf16.txt
cp $O/syncbn_cost.txt $P/r03_syncbn_cost.txt
cp $O/cfg3_autoencoder_optable.txt $P/r03_final_cfg3_optables.txt
cp gpurun_out/pmc_r03f_bf16_wgrad_16_16/summary.txt $P/r03_final_pmc_wgrad_bf16_16_16_summary.txt
for tag in f32_48_16 bf16_48_16 bf16_16_16; do
  cp gpurun_out/pmc_r03f_${tag}/summary.json $P/r03_final_pmc_conv_${tag}_summary.json
  cp gpurun_out/pmc_r03f_${tag}/summary.txt $P/r03_final_pmc_conv_${tag}_summary.txt
done
for dt in f32 bf16; do
  python tools/traffic_from_pmc.py gpurun_out/pmc_r03f_${dt}_48_16 $P/r03_hbm_traffic_48_16.json $dt cat
done
