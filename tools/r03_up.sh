#!/bin/bash
# parity of the resampling kernels, then A/B of the trilinear x2 kernels against the previous build (variant "prev")
set -o pipefail
O=gpurun_out/${1:-r03_up}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "upsampl or resampl or trilinear or pool" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for v in prev default prev default; do
  if [ "$v" = "default" ]; then lib=""; else lib="--lib mri_epilepsy_diagnosis_amd/libmri3d_hip_$v.so"; fi
  echo "[$v]"
  for dt in f32 bf16; do
    timeout -k 10 120 python tools/upsample_probe.py $lib 32 80 96 80 2 $dt 2>/dev/null || exit 1
    timeout -k 10 120 python tools/upsample_probe.py $lib 64 40 48 40 2 $dt 2>/dev/null || exit 1
  done
done | tee $O/ab.txt
