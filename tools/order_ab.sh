#!/bin/bash
# A/B of a tuning build against the shipped library on the three passes of the hot layers, fp32 and bf16.
# usage: tools/order_ab.sh OUT.txt variant [variant...]
out=$1; shift
: > $out
for v in default "$@"; do
  if [ "$v" = "default" ]; then lib=""; else lib="--lib mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so"; fi
  echo "[$v]" >> $out
  for dt in f32 bf16; do
    python tools/conv_bench.py $lib 48 16 160 192 160 2 10 fwd,dgrad,wgrad $dt >> $out 2>/dev/null || exit 1
    python tools/conv_bench.py $lib 16 16 160 192 160 2 10 fwd,wgrad $dt >> $out 2>/dev/null || exit 1
    python tools/conv_bench.py $lib 96 32 80 96 80 2 10 fwd,wgrad $dt >> $out 2>/dev/null || exit 1
  done
done
cat $out
