#!/bin/bash
# SQ / memory counters of one separable-conv pass (tools/sep_conv_bench.py args after the tag), counters only.
#   gpurun -- 'bash tools/pmc_sep.sh TAG <sep_conv_bench args...>'
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sep_$TAG; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq -o pmc -- python3 tools/sep_conv_bench.py "$@" > $O/sq.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY --output-format csv -d $O/sq2 -o pmc -- python3 tools/sep_conv_bench.py "$@" > $O/sq2.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o pmc -- python3 tools/sep_conv_bench.py "$@" > $O/fetch.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "mri3d" in k and "reduce" not in k and "repack" not in k and "pack" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k[:70])
    for c, xs in sorted(v.items()):
        print("    %-26s %14.0f" % (c, sum(xs) / len(xs)))
PY
