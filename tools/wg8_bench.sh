# weight-gradient A/B (default = working tree, prev = previous commit when built as variant "prev")
for v in prev default; do
  if [ "$v" = "default" ]; then lib=""; else lib="--lib mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so"; fi
  [ -z "$lib" ] || [ -f mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so ] || continue
  echo "[$v]"
  python tools/conv_bench.py $lib 8 8 160 192 160 1 20 wgrad 2>/dev/null
  python tools/conv_bench.py $lib 16 8 160 192 160 1 20 wgrad 2>/dev/null
  python tools/conv_bench.py $lib 16 16 160 192 160 2 20 wgrad 2>/dev/null
  python tools/conv_bench.py $lib 32 32 80 96 80 2 20 wgrad 2>/dev/null
  python tools/conv_bench.py $lib --cat 16 48 16 160 192 160 2 20 wgrad 2>/dev/null
  python tools/conv_bench.py $lib --cat 32 96 32 80 96 80 2 20 wgrad 2>/dev/null
  python tools/conv_bench.py $lib 16 16 32 32 32 512 10 wgrad 2>/dev/null
done
for v in prev default; do
  if [ "$v" = "default" ]; then lib=""; else lib="--lib mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so"; fi
  [ -z "$lib" ] || [ -f mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so ] || continue
  echo "[$v bf16]"
  python tools/conv_bench.py $lib 16 16 160 192 160 2 20 wgrad bf16 2>/dev/null
  python tools/conv_bench.py $lib --cat 16 48 16 160 192 160 2 20 wgrad bf16 2>/dev/null
  python tools/conv_bench.py $lib --cat 32 96 32 80 96 80 2 20 wgrad bf16 2>/dev/null
  python tools/conv_bench.py $lib 16 16 32 32 32 512 10 wgrad bf16 2>/dev/null
done
