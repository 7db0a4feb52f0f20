import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, anim_nerf_amd as ana
sys.argv = ["bench.py", "--workload", "cfg4", "--no-extras"]
args = bench.parse()
ctx = bench.Ctx(args)
ana._lib.load()
import torch.distributed as dist
orig = bench.Ctx.timed
def timed(self, step, steps, warmup):
    for _ in range(5): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("synced per step ms: min %.2f med %.2f max %.2f" % (min(ts), sorted(ts)[15], max(ts)))
    t0 = time.perf_counter()
    for _ in range(30): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("unsynced 30 steps: host %.2f ms/step, total %.2f ms/step" % ((t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
    if os.environ.get("PROF"):
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            for _ in range(5): step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25))
    return orig(self, step, steps, warmup)
bench.Ctx.timed = timed
r = bench.train_bench(args, ctx, "bf16", 8, 2)
print(r["ms_per_step"])
