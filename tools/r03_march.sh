#!/bin/bash
# Round-3: parity of the marching kernel, then its A/B against the tiled kernel on the U-Net's layers.
set -o pipefail
O=gpurun_out/${1:-r03_march}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_march_gpu.py -x -q > $O/pytest_march.log 2>&1; rc=$?; echo "pytest march rc=$rc"; tail -15 $O/pytest_march.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/march_bench.py --lib mri_epilepsy_diagnosis_amd/libmri3d_hip_nomarch.so --mode auto --dtype bf16 > $O/ab_tiled_bf16.txt 2>&1; rc=$?; cat $O/ab_tiled_bf16.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/march_bench.py --mode march --dtype bf16 > $O/ab_march_bf16.txt 2>&1; rc=$?; cat $O/ab_march_bf16.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/march_bench.py --lib mri_epilepsy_diagnosis_amd/libmri3d_hip_nomarch.so --mode auto --dtype f32 --layers enc0.conv2,dec1.conv2,enc1.conv1 > $O/ab_tiled_f32.txt 2>&1; rc=$?; cat $O/ab_tiled_f32.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/march_bench.py --mode march --dtype f32 --layers enc0.conv2,dec1.conv2,enc1.conv1 > $O/ab_march_f32.txt 2>&1; rc=$?; cat $O/ab_march_f32.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/host_gap_probe.py m3d 8 nosync > $O/host_gap_nosync.txt 2>&1; rc=$?; tail -20 $O/host_gap_nosync.txt
exit $rc
