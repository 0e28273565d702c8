#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_fuzz_gpu.py tests/test_bf16_gpu.py -x -q -k "bf16 or marching" 2>&1 | tail -2
for v in prev default prev default; do
  if [ "$v" = "default" ]; then lib=""; else lib="--lib mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so"; fi
  echo "[$v bf16]"
  timeout -k 10 120 python tools/conv_bench.py $lib 16 16 160 192 160 2 30 wgrad bf16 2>/dev/null || exit 1
  timeout -k 10 120 python tools/conv_bench.py $lib 8 16 160 192 160 2 30 wgrad bf16 2>/dev/null || exit 1
  timeout -k 10 120 python tools/conv_bench.py $lib --cat 16 48 16 160 192 160 2 30 wgrad bf16 2>/dev/null || exit 1
  timeout -k 10 120 python tools/conv_bench.py $lib --cat 32 96 32 80 96 80 2 30 wgrad bf16 2>/dev/null || exit 1
  timeout -k 10 120 python tools/conv_bench.py $lib 16 16 32 32 32 512 10 wgrad bf16 2>/dev/null || exit 1
done
