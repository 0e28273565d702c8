#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r03_w12}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_march_gpu.py -x -q > $O/pytest_march.log 2>&1; rc=$?; echo "pytest march rc=$rc"; tail -4 $O/pytest_march.log
[ $rc -eq 0 ] || exit $rc
for w in 8 12; do for st in 0 40; do
  echo "== waves $w stagger $st"
  MRI3D_MARCH_WAVES=$w MRI3D_MARCH_STAGGER=$st timeout -k 10 300 python tools/march_bench.py --lib mri_epilepsy_diagnosis_amd/libmri3d_hip_stamps.so --mode march --dtype bf16 --layers dec1.conv2,enc0.conv2,dec1.conv1 2>&1 | grep -v "amdgpu.ids\|^#"
done; done | tee $O/sweep.txt
