# forward / data-gradient A/B of the tiled MFMA kernel (default = working tree, prev = previous commit built as variant "prev")
for v in prev default prev default; do
  if [ "$v" = "default" ]; then lib=""; else lib="--lib mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so"; fi
  echo "[$v]"
  python tools/conv_bench.py $lib 8 16 160 192 160 2 20 fwd,dgrad 2>/dev/null
  python tools/conv_bench.py $lib 16 16 160 192 160 2 20 fwd,dgrad 2>/dev/null
  python tools/conv_bench.py $lib 32 32 80 96 80 2 20 fwd 2>/dev/null
  python tools/conv_bench.py $lib --cat 16 48 16 160 192 160 2 20 fwd,dgrad 2>/dev/null
  python tools/conv_bench.py $lib 16 16 160 192 160 2 20 fwd,dgrad bf16 2>/dev/null
  python tools/conv_bench.py $lib 8 16 160 192 160 2 20 fwd,dgrad bf16 2>/dev/null
  python tools/conv_bench.py $lib --cat 16 48 16 160 192 160 2 20 fwd,dgrad bf16 2>/dev/null
  python tools/conv_bench.py $lib 8 8 160 192 160 1 20 fwd 2>/dev/null
done
