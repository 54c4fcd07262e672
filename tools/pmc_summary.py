#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, per-dispatch mean of every counter."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in rows:
            k = r["Kernel_Name"].split("(")[0][-48:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        print("==", f)
        for k, v in agg.items():
            if "rocclr" in k or "stats_final" in k:
                continue
            n = len(disp[k])
            print(k, "dispatches", n, {a: round(b / n) for a, b in sorted(v.items())})
