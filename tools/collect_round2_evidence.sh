#!/bin/bash
# Copy what tools/profile_round2.sh left under gpurun_out/ into profiles/ (run here, after the gpurun call came back).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof_r02_final
P=profiles
cp $O/final_bench_f32.json $P/r02_final_bench_f32.json
cp $O/final_bench_bf16.json $P/r02_final_bench_bf16.json
grep -v amdgpu.ids $O/final_bench_f32_optable.txt > $P/r02_final_bench_f32_optable.txt
grep -v amdgpu.ids $O/final_bench_bf16_optable.txt > $P/r02_final_bench_bf16_optable.txt
cp $O/bench_f32_under_rocprof.json $P/r02_final_bench_f32_under_rocprof.json
cp $O/bench_bf16_under_rocprof.json $P/r02_final_bench_bf16_under_rocprof.json
cp $O/final_model_bench.txt $P/r02_final_model_bench.txt
cp $O/bench_f32/bench_f32_kernel_stats.csv $P/r02_final_rocprofv3_kernel_stats_bench_f32_steps5.csv
cp $O/bench_bf16/bench_bf16_kernel_stats.csv $P/r02_final_rocprofv3_kernel_stats_bench_bf16_steps5.csv
cp $O/conv_48_16/conv_48_16_kernel_stats.csv $P/r02_final_rocprofv3_kernel_stats_conv_48_16_layer.csv
for dt in f32 bf16; do
  cp gpurun_out/pmc_r02f_${dt}_48_16/summary.json $P/r02_final_pmc_conv_${dt}_48_16_summary.json
  cp gpurun_out/pmc_r02f_${dt}_48_16/summary.txt $P/r02_final_pmc_conv_${dt}_48_16_summary.txt
  python tools/traffic_from_pmc.py gpurun_out/pmc_r02f_${dt}_48_16 $P/r02_hbm_traffic_48_16.json $dt cat
done
