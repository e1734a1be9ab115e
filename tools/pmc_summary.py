#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs (mean per dispatch, per kernel) from a directory."""
import collections, csv, glob, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "asrfe" not in k:
            continue
        agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for n, v in sorted(c.items()):
        print(f"   {n:28s} n={len(v):4d} mean={sum(v)/len(v):.5g}")
