#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of bench.py (fp32 headline and bf16), summaries under gpurun_out/prof/
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (never combined with other trace domains) on the dominant
#      48->16 layer, one conv pass per profiler run so that forward and data-gradient (same kernel) can be told apart.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_f32 -o bench_f32 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err || exit 1
echo "bench f32 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_bf16 -o bench_bf16 -- python3 bench.py --steps 5 --warmup 2 --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err || exit 1
echo "bench bf16 done"
# per-layer kernel durations of the dominant 48->16 layer (the bench-level stats pool every layer served by one kernel name)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/conv_48_16 -o conv_48_16 -- python3 tools/conv_bench.py 48 16 160 192 160 2 10 fwd,dgrad,wgrad > $O/conv_48_16.log 2>&1 || exit 1
echo "conv 48->16 stats done"
for pass in fwd dgrad wgrad; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${pass}_${ctr} -o pmc -- python3 tools/conv_bench.py 48 16 160 192 160 2 3 $pass > $O/pmc_${pass}_${ctr}.log 2>&1 || exit 1
  done
  echo "pmc $pass done"
done
find $O -name "*.csv" | head -40
