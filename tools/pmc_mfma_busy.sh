#!/bin/bash
# MFMA-pipe utilisation of the dominant 48->16 layer's three passes: SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE
# (counters only: --pmc is never combined with trace domains).   gpurun -- 'bash tools/pmc_mfma_busy.sh'
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_busy
mkdir -p $O
cd $R
for pass in fwd dgrad wgrad; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/$pass -o pmc -- python3 tools/conv_bench.py 48 16 160 192 160 2 3 $pass > $O/$pass.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --output-format csv -d $O/${pass}_insts -o pmc -- python3 tools/conv_bench.py 48 16 160 192 160 2 3 $pass > $O/${pass}_insts.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
out = []
for p in ("fwd", "dgrad", "wgrad"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in (p, p + "_insts"):
        for f in glob.glob("gpurun_out/pmc_busy/%s/**/*counter_collection.csv" % d, recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "mfma" in k and "pack" not in k and "reduce" not in k:
                    agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        line = "%-6s %-46s" % (p, k[:46])
        if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs, the MFMA counter over the 1024 SIMDs
            line += " MFMA busy %.1f %%" % (100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))
        line += "  " + "  ".join("%s=%.4g" % (c, m[c]) for c in sorted(m))
        out.append(line)
print("\n".join(out))
open("gpurun_out/pmc_busy/summary.txt", "w").write("\n".join(out) + "\n")
PY
