#!/bin/bash
# Round-3 opening run on the GPU box: full GPU suite (ordinary interpreter exit, faulthandler file), headline bench with the
# `secondary` object, bf16 bench, the other configurations' operator tables, and the host-gap probe of the eager Modified3DUNet.
set -o pipefail
O=gpurun_out/${1:-r03_base}
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee $O/pytest.rc; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
python bench.py > $O/bench_f32.json 2> $O/bench_f32.err; rc=$?; echo "bench f32 rc=$rc"; cat $O/bench_f32.json
[ $rc -eq 0 ] || exit $rc
python bench.py --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err; rc=$?; echo "bench bf16 rc=$rc"; cat $O/bench_bf16.json
[ $rc -eq 0 ] || exit $rc
python tools/host_gap_probe.py m3d 3 > $O/host_gap_m3d.txt 2>&1; rc=$?; echo "probe rc=$rc"
[ $rc -eq 0 ] || exit $rc
python tools/model_bench.py all > $O/model_bench.txt 2>&1; rc=$?; echo "model_bench rc=$rc"
exit $rc
