#!/usr/bin/env python3
"""Per-kernel HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected separately,
MI355X_MICROARCH.md "HBM"): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE counts 128-byte
requests as 64 bytes for wide coalesced reads, hence the factor 2.  Usage: pmc_traffic.py DIR OUT.json [substr]"""
import collections, csv, glob, json, sys
csv.field_size_limit(1 << 30)
pat = sys.argv[3] if len(sys.argv) > 3 else "advgrpo"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            a = acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
out = {}
for k, c in acc.items():
    fe = c["FETCH_SIZE"][0] / max(1, c["FETCH_SIZE"][1])
    wr = c["WRITE_SIZE"][0] / max(1, c["WRITE_SIZE"][1])
    out[k] = {"launches": c["FETCH_SIZE"][1], "FETCH_SIZE_kb": fe, "WRITE_SIZE_kb": wr,
              "hbm_bytes_per_launch": (2 * fe + wr) * 1024}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k[:80]:80s} n={v['launches']:5d} fetch {v['FETCH_SIZE_kb'] / 1024:9.1f} MB(raw) write {v['WRITE_SIZE_kb'] / 1024:9.1f} MB  hbm {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB")
