#!/usr/bin/env python3
"""The launches of ONE replayed training step, in order, from a rocprofv3 kernel trace of `bench.py --workload cfg4` (steps are
delimited by the Adam launch): duration, start offset, kernel — the evidence that no framework kernel sits between the step's
first and last launch.   python tools/step_sequence.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
mid = len(ends) // 2                                          # (a replayed step: the run ends with a few eager ones)
a, b = ends[mid], ends[mid + 1]
seg = rows[a + 1:b + 1]
ours = sum(1 for _, _, n in seg if "anr::" in n or "_ZN3anr" in n)
print(f"{len(seg)} launches in the step, {ours} of them kernels of libanimnerf_hip.so; others: "
      f"{sorted({n.split('(')[0][:60] for _, _, n in seg if not ('anr::' in n or '_ZN3anr' in n)})}")
print("   us     at us   kernel")
for s, e, n in seg:
    print(f"{(e - s) / 1e3:7.1f} {(s - rows[a][1]) / 1e3:9.1f}   {n.split('(')[0][:100]}")
