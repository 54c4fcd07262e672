#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (csv or rocpd sqlite):
  --pmc runs: per kernel, per-dispatch mean of every counter;
  --kernel-trace --stats runs: the per-kernel duration table (calls, total us, average us, %)."""
import collections
import csv
import glob
import sqlite3
import sys


def short(name):
    return name.split("(")[0][-48:]


def counters_csv(f):
    for r in csv.DictReader(open(f)):
        yield r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"]), r["Dispatch_Id"]


def counters_db(f):
    db = sqlite3.connect(f)
    try:
        for r in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
            yield r
    except sqlite3.Error:
        return


def stats_db(f):
    db = sqlite3.connect(f)
    try:
        return list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    except sqlite3.Error:
        return []


for d in sys.argv[1:]:
    srcs = [(f, counters_csv(f)) for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True)]
    dbs = glob.glob(d + "/**/*results.db", recursive=True)
    srcs += [(f, counters_db(f)) for f in dbs]
    for f, rows in srcs:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for kname, cname, val, did in rows:
            k = short(kname)
            agg[k][cname] += float(val)
            disp[k].add(did)
        if not agg:
            continue
        print("==", f)
        for k, v in agg.items():
            if "rocclr" in k or "stats_final" in k:
                continue
            n = len(disp[k])
            print(k, "dispatches", n, {a: round(b / n) for a, b in sorted(v.items())})
    for f in dbs:
        rows = stats_db(f)
        if rows and not any(True for _ in counters_db(f)):
            print("==", f)
            print('"Name","Calls","TotalDurationUs","AverageUs","Percentage"')
            for name, calls, tot, avg, pct in rows:
                print(f'"{name}",{calls},{tot:.3f},{avg:.3f},{pct:.2f}')
