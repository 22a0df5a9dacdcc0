#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per launch for kernels matching a substring."""
import collections, csv, glob, sys
pat, dirs = sys.argv[1], sys.argv[2:]
for d in dirs:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(f"{k:32s} n={len(v)} mean={sum(v)/len(v):.4g}")
    for f in sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[:1]:
        ds = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
        if ds:
            print(f"duration_ns mean={sum(ds)/len(ds):.0f} n={len(ds)}")
