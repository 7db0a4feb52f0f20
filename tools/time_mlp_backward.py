import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import autograd as ag
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ana.NeRF(freqs_dir=0, use_view=False, mlp_mode="bf16").to(dev)
n = 1048576
pts = torch.cat([torch.rand(n, 3, device=dev) * 2 - 1, torch.ones(n, 1, device=dev)], -1)
g = torch.randn(n, 4, device=dev)
import torch.utils.benchmark as tb
for it in range(4):
    net.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = net.eval_points(pts)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    (out * g).sum().backward()
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"fwd {1e3*(t1-t0):.1f} ms   bwd cpu {1e3*(t2-t1):.1f} ms   bwd total {1e3*(t3-t1):.1f} ms")
# instrument pieces
orig_bmm = torch.bmm
acc = {"bmm": 0.0, "n": 0}
def timed_bmm(*a, **k):
    t = time.perf_counter(); r = orig_bmm(*a, **k); acc["bmm"] += time.perf_counter() - t; acc["n"] += 1; return r
torch.bmm = timed_bmm
net.zero_grad(set_to_none=True)
out = net.eval_points(pts); torch.cuda.synchronize()
(out * g).sum().backward(); torch.cuda.synchronize()
print("bmm cpu total ms", 1e3 * acc["bmm"], acc["n"])
rows = []
def timed_bmm2(a, b, **k):
    t = time.perf_counter(); r = orig_bmm(a, b, **k); dt = time.perf_counter() - t
    rows.append((round(1e3 * dt, 3), tuple(a.shape), a.stride(), tuple(b.shape), b.stride(), k)); return r
torch.bmm = timed_bmm2
net.zero_grad(set_to_none=True)
out = net.eval_points(pts); torch.cuda.synchronize()
(out * g).sum().backward(); torch.cuda.synchronize()
for r in rows: print(r)
