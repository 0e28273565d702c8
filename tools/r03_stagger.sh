#!/bin/bash
# stagger sweep of the marching kernel (tuning build reads MRI3D_MARCH_STAGGER)
for st in 0 20 40 60 80; do
  echo "== stagger $st"
  MRI3D_MARCH_STAGGER=$st timeout -k 10 300 python tools/march_bench.py --lib mri_epilepsy_diagnosis_amd/libmri3d_hip_stamps.so --mode march --dtype bf16 --layers dec1.conv2,dec1.conv1 2>&1 | grep -v "amdgpu.ids\|^#"
done
