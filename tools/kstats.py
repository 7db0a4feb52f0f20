#!/usr/bin/env python3
"""Per-frame kernel time table from a rocprofv3 --kernel-trace --stats csv directory.  Usage: kstats.py <dir> <frames>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in list(csv.DictReader(open(f)))[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} {float(r['TotalDurationNs'])/frames/1e6:8.3f} ms/frame avg {float(r['AverageNs'])/1e6:7.3f} ms")
