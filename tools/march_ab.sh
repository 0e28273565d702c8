# weight gradient marching along d (default) against the tile kernels (variant "tilewg":
#   python -m mri_epilepsy_diagnosis_amd.build --variant tilewg -DMRI3D_F32_WGRAD_MARCH_MIN_D=1000000 -DMRI3D_BF16_WGRAD_MARCH_MIN_D=1000000)
DT=${1:-f32}
for v in tilewg default; do
  if [ "$v" = "default" ]; then lib=""; else lib="--lib mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so"; fi
  echo "[$v $DT]"
  python tools/conv_bench.py $lib 16 16 160 192 160 2 20 wgrad $DT 2>/dev/null
  python tools/conv_bench.py $lib --cat 16 48 16 160 192 160 2 20 wgrad $DT 2>/dev/null
  python tools/conv_bench.py $lib --cat 32 96 32 80 96 80 2 20 wgrad $DT 2>/dev/null
  python tools/conv_bench.py $lib 32 32 80 96 80 2 20 wgrad $DT 2>/dev/null
  python tools/conv_bench.py $lib 16 32 80 96 80 2 20 wgrad $DT 2>/dev/null
  python tools/conv_bench.py $lib 32 64 40 48 40 2 20 wgrad $DT 2>/dev/null
  python tools/conv_bench.py $lib 16 16 32 32 32 512 10 wgrad $DT 2>/dev/null
done
