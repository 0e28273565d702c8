#!/bin/bash
for w in 8 12; do
  echo "== waves $w, no stores"
  MRI3D_MARCH_WAVES=$w timeout -k 10 300 python tools/march_bench.py --lib mri_epilepsy_diagnosis_amd/libmri3d_hip_nostore.so --mode march --dtype bf16 --layers dec1.conv2 2>&1 | grep -v "amdgpu.ids\|^#"
done
