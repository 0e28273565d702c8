#!/bin/bash
# rocprofv3 kernel trace of the marching kernel on the U-Net's bf16 layers (kernel durations without launch gaps)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_ktrace}
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 tools/march_bench.py --mode march --dtype bf16 --layers dec1.conv2,dec1.conv1,enc0.conv2 --reps 20 > $O/run.log 2>&1
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -12 $O/kernel_stats.csv | cut -c1-200
grep -v amdgpu.ids $O/run.log | tail -8
