#!/bin/bash
# rocprofv3 kernel stats of the cfg3 encoder + head step replayed as a hipGraph
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_cfg3
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -o enc -- python3 tools/model_bench.py cfg3_graph > $O/enc.log 2>&1 || exit 1
python3 - <<'PY'
import csv, os
f = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/prof_cfg3/enc/enc_kernel_stats.csv")
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels total %.3f ms over %d kernel names" % (tot / 1e6, len(rows)))
for r in rows[:45]:
    print("%-100s %5s %9.1f us avg %8.2f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
