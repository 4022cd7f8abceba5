#!/usr/bin/env python3
"""Per-kernel mean of each PMC counter from rocprofv3 --pmc CSV output.  Usage: pmc_csv.py DIR [name-substring]"""
import csv, glob, sys, collections
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            k = (row["Kernel_Name"][:60], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (v, n) in sorted(acc.items()):
    print(f"{kn:60s} {cn:28s} mean {v / n:16.1f}  n={n}")
