#!/bin/bash
timeout -k 10 300 python tools/march_bench.py --mode march --dtype f32 --reps 10 --layers enc0.conv2,dec1.conv2,dec1.conv1,dec0.conv1,dec0.conv2,enc1.conv2 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/march_bench.py --mode auto --dtype f32 --reps 10 --layers enc0.conv2,dec1.conv2,dec1.conv1,dec0.conv1,dec0.conv2,enc1.conv2 2>&1 | grep -v amdgpu.ids
