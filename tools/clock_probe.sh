#!/bin/bash
# sample GPU clocks while a conv kernel loops
(timeout -k 5 90 python tools/conv_bench.py 48 16 160 192 160 2 8000 fwd > gpurun_out/clk_bench.log 2>&1) &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -6
  echo "--"
  sleep 1
done
wait $BP
cat gpurun_out/clk_bench.log | grep -v amdgpu
