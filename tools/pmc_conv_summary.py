"""Summarise tools/pmc_conv.sh's counter CSVs: per conv pass, per-launch averages of the MFMA kernel's counters, the algorithmic
bytes / FLOPs of the pass, and the derived ratios (fabric bytes fetched per algorithmic byte, MFMA-pipe busy share)."""
import collections
import csv
import glob
import json
import os
import sys

out_dir, ci, co, d, h, w, n, dt = sys.argv[1], *map(int, sys.argv[2:8]), sys.argv[8]
passes = sys.argv[9].split()
es = 2 if dt == "bf16" else 4
vox = n * d * h * w
alg = {"fwd": vox * (ci + co) * es, "dgrad": vox * (ci + co) * es, "wgrad": vox * (ci + co) * es}
flops = 2.0 * vox * ci * co * 27
lines, rec = [], {}
for p in passes:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("fetch", "write", "sq"):
        for f in glob.glob(os.path.join(out_dir, "%s_%s" % (p, sub), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if ("mfma" in k or "conv_" in k) and "pack" not in k and "reduce" not in k:
                    agg[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        e = {"kernel": k, "counters_per_launch": m, "algorithmic_bytes": alg[p], "flops": flops}
        if "FETCH_SIZE" in m:
            e["raw_fetch_bytes"] = 1024.0 * m["FETCH_SIZE"]
        if "WRITE_SIZE" in m:
            e["write_bytes"] = 1024.0 * m["WRITE_SIZE"]
        if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if "SQ_LDS_IDX_ACTIVE" in m and "GRBM_GUI_ACTIVE" in m:
            e["lds_active_frac_per_cu"] = m["SQ_LDS_IDX_ACTIVE"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256.0)
        rec["%s %s" % (p, k[:60])] = e
        lines.append("%-6s %-44s raw fetch %8.1f MB  write %8.1f MB  (algorithmic %8.1f MB)  MFMA busy %5.1f %%  LDS active %5.1f %%  conflicts/active %.3f"
                     % (p, k[:44], e.get("raw_fetch_bytes", 0) / 1e6, e.get("write_bytes", 0) / 1e6, alg[p] / 1e6,
                        100 * e.get("mfma_busy_frac", 0), 100 * e.get("lds_active_frac_per_cu", 0),
                        m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
print("\n".join(lines))
open(os.path.join(out_dir, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(rec, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
