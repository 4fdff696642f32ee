#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection.csv files per kernel and counter: tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in sorted(glob.glob(d + "/pass*/*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if filt and filt not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
for k in acc:
    print("==", k)
    for c, v in sorted(acc[k].items()):
        print("  %-28s %18.0f   (%d dispatches)" % (c, v, len(calls[(k, c)])))
