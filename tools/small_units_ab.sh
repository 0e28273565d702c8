for v in default su512 su1024; do
  if [ "$v" = "default" ]; then lib=""; else lib="--lib mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so"; fi
  echo "[$v]"
  python tools/conv_bench.py $lib 32 32 40 48 40 2 20 fwd,dgrad 2>/dev/null
  python tools/conv_bench.py $lib 32 64 40 48 40 2 20 fwd,dgrad 2>/dev/null
  python tools/conv_bench.py $lib 64 64 40 48 40 1 20 fwd,dgrad 2>/dev/null
  python tools/conv_bench.py $lib 32 32 40 48 40 1 20 fwd 2>/dev/null
done
