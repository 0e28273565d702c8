#!/bin/bash
# when do the workgroups of the marching kernel start and end (needs the tuning build --variant stamps -DMRI3D_EXPERIMENT_STAMPS)
timeout -k 10 300 python tools/march_bench.py --lib mri_epilepsy_diagnosis_amd/libmri3d_hip_stamps.so --mode march --dtype bf16 --layers dec1.conv2,dec1.conv1 2>&1 | grep -v "amdgpu.ids\|^#"
