#!/bin/bash
# HBM traffic of the trilinear x2 kernels on the decoder's 32-channel level (coarse 80x96x80, n2): FETCH_SIZE / WRITE_SIZE in
# separate rocprofv3 passes.   gpurun -- 'bash tools/pmc_upsample.sh'
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_up
mkdir -p $O
cd $R
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $O/$ctr -o pmc -- python3 tools/upsample_probe.py 32 80 96 80 2 f32 > $O/$ctr.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_up/%s/**/*counter_collection.csv" % ctr, recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "upsample" in k:
            print(ctr, k, "launches", len(v), "avg", sum(v) / len(v))
PY
