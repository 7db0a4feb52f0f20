#!/usr/bin/env python3
"""Timeline of one replayed training step from a rocprofv3 kernel trace (bench.py --workload cfg4): start, duration and queue of every
launch, how many launches run at the same time, and the time the GPU runs nothing.   python tools/exp/step_timeline.py <trace dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'train_draws_kernel' in r['Kernel_Name']]
i0, i1 = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
step = rows[i0:i1]
t0 = int(step[0]['Start_Timestamp'])
end = max(int(r['End_Timestamp']) for r in step)
print(f"one step: {len(step)} launches, wall {(end - t0) / 1e3:.1f} us, sum of kernel times {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3:.1f} us")
ev = sorted([(int(r['Start_Timestamp']), 1) for r in step] + [(int(r['End_Timestamp']), -1) for r in step])
busy = {}; cur = 0; last = t0
for t, d in ev:
    busy[cur] = busy.get(cur, 0) + (t - last); last = t; cur += d
print("time with k launches running:", {k: round(v / 1e3, 1) for k, v in sorted(busy.items())}, "us")
q = {}
for r in step:
    qid = r.get('Queue_Id', '?')
    q.setdefault(qid, len(q))
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  q{q[qid]}  {r['Kernel_Name'][:90]}")
