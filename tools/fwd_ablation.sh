#!/bin/bash
# A/B of tuning builds of the forward kernel (libmri3d_hip_<variant>.so from `python -m mri_epilepsy_diagnosis_amd.build --variant`).
# usage: tools/fwd_ablation.sh OUT.txt variant [variant ...]      ("default" = the shipped library)
out=$1; shift
: > $out
for v in "$@"; do
  if [ "$v" = "default" ]; then lib=""; else lib="--lib mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so"; fi
  echo "[$v]" >> $out
  python tools/conv_bench.py $lib 48 16 160 192 160 2 10 fwd,dgrad >> $out 2>/dev/null || exit 1
  python tools/conv_bench.py $lib 16 16 160 192 160 2 10 fwd >> $out 2>/dev/null || exit 1
  python tools/conv_bench.py $lib 48 16 160 192 160 2 10 fwd,dgrad bf16 >> $out 2>/dev/null || exit 1
  python tools/conv_bench.py $lib 16 16 160 192 160 2 10 fwd bf16 >> $out 2>/dev/null || exit 1
  python tools/conv_bench.py $lib 32 32 80 96 80 2 10 fwd >> $out 2>/dev/null || exit 1
  python tools/conv_bench.py $lib 32 32 80 96 80 2 10 fwd bf16 >> $out 2>/dev/null || exit 1
done
cat $out
