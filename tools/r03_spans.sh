#!/bin/bash
for w in 8 12; do
  echo "== waves $w"
  MRI3D_MARCH_WAVES=$w timeout -k 10 300 python tools/march_bench.py --lib mri_epilepsy_diagnosis_amd/libmri3d_hip_stamps.so --mode march --dtype bf16 --layers dec1.conv2,dec1.conv1 2>&1 | grep -v "amdgpu.ids\|^#"
done
