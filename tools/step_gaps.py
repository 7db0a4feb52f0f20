#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `bench.py --workload cfg4`: per training step (delimited by the Adam launches),
the step's wall time on the GPU, the sum of its kernel durations, the time with NO kernel running (the step's launches overlap
since round 4: parallel branches of its graph) and the time with k kernels running; the median step, and its kernels ranked by time.   python tools/step_gaps.py <dir with *_kernel_trace.csv>"""
import collections
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2] or "FusedOptimizerTensorListMetadata" in r[2]]
# a step has a few Adam launches back to back: keep the last of each cluster
last = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] - i > 5]
steps = []
for a, b in zip(last[:-1], last[1:]):
    seg = rows[a + 1:b + 1]
    t0 = rows[a][1]
    wall = max(e for _, e, _ in seg) - t0
    busy = sum(e - s for s, e, _ in seg)
    ev = sorted([(max(s, t0), 1) for s, e, _ in seg] + [(e, -1) for s, e, _ in seg])
    conc, cur, at = collections.Counter(), 0, t0
    for t, d in ev:
        conc[cur] += t - at
        at, cur = t, cur + d
    steps.append((wall, busy, len(seg), seg, conc))
steps.sort(key=lambda t: t[0])
print("steps found", len(steps))
for q in (0.1, 0.5, 0.9):
    w, b, n, _, conc = steps[int(q * (len(steps) - 1))]
    print(f"  q{int(q * 100):02d}: wall {w / 1e6:.3f} ms   sum of kernel times {b / 1e6:.3f} ms   nothing running {conc[0] / 1e6:.3f} ms   launches {n}   "
          f"with k running: " + ", ".join(f"{k}: {v / 1e6:.2f}" for k, v in sorted(conc.items()) if k))
w, b, n, seg, _ = steps[len(steps) // 2]
acc = collections.defaultdict(lambda: [0, 0])
for s, e, name in seg:
    acc[name.split("(")[0][:80]][0] += e - s
    acc[name.split("(")[0][:80]][1] += 1
for name, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"  {t / 1e3:8.1f} us {c:4d}x  {name}")
