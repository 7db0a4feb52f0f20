#!/usr/bin/env python3
"""Experiment (round 5): does an EXTERNAL event recorded in the middle of a captured HIP graph (torch.cuda.Event(external=True):
an event-record NODE) let a stream outside the graph start work while the rest of the replay is still running?  That is what
an all-reduce of the fine network's gradient bucket overlapping the rest of a replayed backward pass needs."""
import time
import torch
dev = torch.device("cuda:0")
main, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
x = torch.zeros(1 << 20, device=dev)
src = torch.zeros(1 << 20, device=dev)
big = torch.randn(4096, 4096, device=dev)
out = torch.empty_like(big)
y = torch.zeros(1 << 20, device=dev)
ev = torch.cuda.Event(external=True)
done_side = torch.cuda.Event(enable_timing=True)
t0e, t1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    for _ in range(3):
        x.copy_(src); torch.mm(big, big, out=out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=main):
        x.copy_(src)                       # "the fine bucket is complete"
        ev.record(main)                    # event-record node
        for _ in range(20):                # "the rest of the backward pass" (~ms)
            torch.mm(big, big, out=out)
ok = True
for it in range(1, 6):
    src.fill_(float(it))
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        t0e.record(main)
        g.replay()
        t1e.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        y.copy_(x)
        done_side.record(side)
    torch.cuda.synchronize()
    good = bool((y == float(it)).all())
    ok &= good
    print(f"replay {it}: side stream saw the value of THIS replay: {good}; graph {t0e.elapsed_time(t1e):.2f} ms, side work done "
          f"{t0e.elapsed_time(done_side):.2f} ms after the replay began")
print("external event in a captured graph:", "works" if ok else "BROKEN")
