#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc csv directory: per kernel-name prefix, sum of each counter and launch count.
Usage: python tools/pmc_summary.py <dir> [name-substring ...]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
subs = sys.argv[2:]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in files:
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if subs and not any(s in name for s in subs):
            continue
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        launches[name].add(row["Dispatch_Id"])
for name in sorted(acc):
    print(name[:110], "launches", len(launches[name]))
    for c, v in sorted(acc[name].items()):
        print(f"    {c:32s} {v:.4g}")
