#!/usr/bin/env python3
"""Per-call duration of the warp kernels (coarse call, fine call) from a kernel trace directory of bench.py --workload cfg3."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
for k in ['warp_search_kernel', 'warp_cells_kernel', 'warp_classify']:
    print(f"{k:22s}", [round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if k in r['Kernel_Name']], "us")
