#!/bin/bash
# full GPU suite + both bench lines:  gpurun --timeout 1100 -- "bash tools/gpu_suite_and_bench.sh TAG"  ->  gpurun_out/TAG/
set -o pipefail
O=gpurun_out/${1:-suite}
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee $O/pytest.rc; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
python bench.py --no-secondary > $O/bench_f32.json 2> $O/bench_f32.err; rc=$?; echo "bench f32 rc=$rc"; cut -c1-400 $O/bench_f32.json
[ $rc -eq 0 ] || exit $rc
python bench.py --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err; rc=$?; echo "bench bf16 rc=$rc"; cut -c1-400 $O/bench_bf16.json; grep -v amdgpu.ids $O/bench_bf16.err | head -8
exit $rc
