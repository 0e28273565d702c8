"""profiles/r01_final_pmc_48_16_<pass>_<COUNTER>_counter_collection.csv  ->  profiles/r01_hbm_traffic_48_16.json
(per-launch HBM traffic of the three passes of the dominant 48->16 layer, keyed by bench.py's operator tags).

Counters come from SEPARATE rocprofv3 --pmc passes (tools/profile_round.sh), are in KiB, and are averaged over the
launches of the MFMA kernel in that run.  Correction, calibrated on this path's own pattern as MI355X_MICROARCH.md asks:
the x2 FETCH_SIZE correction holds for wide coalesced streams (the weight-gradient reduce kernel reads a known 14.6 MB
and counts 7.35 MB raw), but the MFMA kernels stage 32-byte halo pieces at a 192-byte voxel stride — 64-byte requests
that the counter tallies exactly (16->16 forward: raw 1.18 GB against 0.63 GB unique + halo re-reads) — so for them
traffic = raw FETCH + WRITE.  Both figures are stored."""
import collections
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = "conv3d_%s 3x3x3 s1 d1 48->16 @160x192x160 n2"


def per_launch(path, counter):
    tot, ids, name = 0.0, set(), None
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] != counter or "mfma" not in k or "pack" in k or "reduce" in k:
            continue
        tot += float(r["Counter_Value"])
        ids.add(r["Dispatch_Id"])
        name = k.split("(")[0]
    return 1024.0 * tot / len(ids), name, len(ids)


out = {"_note": __doc__.split("\n\n", 1)[1].replace("\n", " ")}
for p in ("fwd", "dgrad", "wgrad"):
    f, name, n = per_launch(os.path.join(P, "r01_final_pmc_48_16_%s_FETCH_SIZE_counter_collection.csv" % p), "FETCH_SIZE")
    w, _, _ = per_launch(os.path.join(P, "r01_final_pmc_48_16_%s_WRITE_SIZE_counter_collection.csv" % p), "WRITE_SIZE")
    out[TAG % p] = {"kernel": name, "launches_averaged": n, "raw_fetch_bytes": f, "write_bytes": w, "traffic_bytes": f + w,
                    "fetch_bytes_if_x2_correction": 2 * f}
    print("%-6s %-50s fetch %.3f GB  write %.3f GB" % (p, name, f / 1e9, w / 1e9))
json.dump(out, open(os.path.join(P, "r01_hbm_traffic_48_16.json"), "w"), indent=1)
