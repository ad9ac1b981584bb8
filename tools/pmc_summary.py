#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc csv passes: per kernel (substring match) the per-dispatch mean of every counter.
usage: tools/pmc_summary.py <dir with 1/,2/,... passes> <kernel substring>"""
import csv, glob, sys, collections
csv.field_size_limit(1 << 30)
root, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in sorted(set(glob.glob(root + "/**/*counter_collection.csv", recursive=True))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k:
            continue
        k = k.split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, r["Counter_Name"])].add((f, r["Dispatch_Id"]))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        n = len(disp[(k, c)])
        print(f"  {c:28s} {v / n:18.0f}   (mean of {n} dispatches)")
