for v in default sametile nostage nowload nomem nobar; do
  if [ "$v" = "default" ]; then lib=""; else lib="--lib mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so"; fi
  echo "[$v]"
  python tools/conv_bench.py $lib 8 16 160 192 160 2 10 fwd 2>/dev/null
  python tools/conv_bench.py $lib 16 16 160 192 160 2 10 fwd 2>/dev/null
  python tools/conv_bench.py $lib 16 16 160 192 160 2 10 fwd bf16 2>/dev/null
done
