"""Parse rocprofv3 --pmc CSVs (FETCH_SIZE and WRITE_SIZE collected in two separate passes, as MI355X_MICROARCH.md
§HBM prescribes) into per-launch HBM traffic per kernel, applying the gfx950 correction (FETCH_SIZE reports half of a
wide coalesced read stream; counters are in KiB).

    python tools/collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    tot, calls = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    return {k: (tot[k], len(calls[k])) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 1))
        w, nw = write.get(k, (0.0, 1))
        # counters are in KiB; FETCH_SIZE x2 on gfx950 for 16-byte-per-lane streams (guide's correction)
        out[k] = {"launches": max(nf, nw), "fetch_bytes_per_launch": 2.0 * 1024.0 * f / max(nf, 1),
                  "write_bytes_per_launch": 1024.0 * w / max(nw, 1),
                  "raw_fetch_kib_per_launch": f / max(nf, 1), "raw_write_kib_per_launch": w / max(nw, 1)}
        out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        if "mfma" in k or "norm" in k:
            print("%-60s %8.1f MB fetch(x2) %8.1f MB write per launch" % (k[:60], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
