"""gpurun_out/pmc_<tag>/summary.json (tools/pmc_conv.sh)  ->  profiles/<round>_hbm_traffic_48_16.json, keyed by bench.py's operator tags.

Correction (MI355X_MICROARCH.md §HBM, calibrated this round on this library's own access shapes with
tools/microbench/fetch_calib.hip, profiles/r02_fetch_size_calibration.txt): FETCH_SIZE tallies 64 bytes per 128-byte line
request for EVERY pattern tried — a 16-byte-per-lane stream, and 32-byte pieces at 32 / 64 / 96 / 192-byte voxel pitch (the halo
staging of the conv kernels) all report exactly half of the bytes of the lines they touch.  So fetched bytes = 2 x FETCH_SIZE
(round 1 took the raw value for the halo pieces, which under-counted by 2x); WRITE_SIZE is taken as reported.  The figure is
memory-side (fabric) traffic of the XCD L2s: Infinity-Cache hits are included."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, out, dt = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "f32")
cat = " cat" if (len(sys.argv) > 4 and sys.argv[4] == "cat") else ""     # the split-operand passes (tools/pmc_conv.sh ... SPLIT)
d = json.load(open(os.path.join(src, "summary.json")))
suffix = " bf16" if dt == "bf16" else ""
rec = {"_note": __doc__.split("\n\n", 1)[1].replace("\n", " "), "_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_conv.sh"}
for key, e in d.items():
    p = key.split()[0]
    tag = "conv3d_%s 3x3x3 s1 d1 48->16 @160x192x160 n2%s%s" % (p, suffix, cat)
    rec[tag] = {"kernel": e["kernel"], "raw_fetch_size_bytes": e.get("raw_fetch_bytes"), "fetch_bytes": 2.0 * e.get("raw_fetch_bytes", 0.0),
                "write_bytes": e.get("write_bytes"), "traffic_bytes": 2.0 * e.get("raw_fetch_bytes", 0.0) + e.get("write_bytes", 0.0),
                "algorithmic_bytes": e["algorithmic_bytes"], "mfma_busy_frac": e.get("mfma_busy_frac")}
    print("%-6s fetch %.2f GB  write %.2f GB  traffic %.2f GB  (algorithmic %.2f GB)" % (p, rec[tag]["fetch_bytes"] / 1e9, rec[tag]["write_bytes"] / 1e9, rec[tag]["traffic_bytes"] / 1e9, e["algorithmic_bytes"] / 1e9))
prev = json.load(open(out)) if os.path.exists(out) else {}
prev.update(rec)
json.dump(prev, open(out, "w"), indent=1)
