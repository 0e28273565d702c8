#!/bin/bash
# parity of the marching kernel + stamps + A/B (bf16), one call
set -o pipefail
O=gpurun_out/${1:-r03_march2}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_march_gpu.py -x -q > $O/pytest_march.log 2>&1; rc=$?; echo "pytest march rc=$rc"; tail -4 $O/pytest_march.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/march_bench.py --lib mri_epilepsy_diagnosis_amd/libmri3d_hip_stamps.so --mode march --dtype bf16 --layers dec1.conv2,dec1.conv1,enc0.conv2 2>&1 | grep -v amdgpu.ids | tee $O/stamps_bf16.txt
timeout -k 10 300 python tools/march_bench.py --mode march --dtype bf16 --layers ${2:-enc0.conv2,dec1.conv2,dec1.conv1,enc1.conv1,dec0.conv2} 2>&1 | grep -v amdgpu.ids | tee $O/ab_march_bf16.txt
if [ "$3" = "tiled" ]; then
timeout -k 10 300 python tools/march_bench.py --lib mri_epilepsy_diagnosis_amd/libmri3d_hip_nomarch.so --mode auto --dtype bf16 --layers ${2:-enc0.conv2,dec1.conv2,dec1.conv1,enc1.conv1,dec0.conv2} 2>&1 | grep -v amdgpu.ids | tee $O/ab_tiled_bf16.txt
fi
exit 0
