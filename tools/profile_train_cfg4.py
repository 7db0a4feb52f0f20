#!/usr/bin/env python3
"""torch.profiler of one cfg4 training step (as bench.py --workload cfg4): GEMM-like ops grouped by input shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
torch.manual_seed(0)
model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True, use_fine=True, mlp_mode="bf16").to(dev)
hp = ana.TrainHParams(n_samples=64, n_importance=32, chunk=2048)
F = 16
table = ana.BodyModelParams(114).to(dev)
seeded = syn.animated_pose_params(seed=200, bs=114)
for name in table.param_names:
    table.init_parameters(name, torch.from_numpy(seeded[name]).to(dev), requires_grad=True)
tr = ana.Trainer(model, ana.VolumeRenderer(n_coarse=64, n_fine=32), hp, body_model_params=table)
frame_idx = torch.arange(F, device=dev) * (114 // F)
c2w, focal, cen = syn.pinhole_camera(32, 32)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 32, 32, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(F, 1, 1, 1)
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
g = torch.Generator().manual_seed(0)
rgbs = torch.rand(F, 32, 32, 3, generator=g).to(dev); alphas = (torch.rand(F, 32, 32, 1, generator=g) > 0.5).float().to(dev)
fg = (torch.rand(F, 128, 3, generator=g) * 0.4 - 0.2).to(dev); bg = (torch.rand(F, 128, 3, generator=g) * 2 - 1).to(dev)
step = lambda: tr.step(rays, rgbs, alphas, None, templ, fg, bg, perturb=1.0, frame_idx=frame_idx)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::mm", "aten::bmm", "aten::addmm", "aten::matmul", "aten::linear", "aten::einsum", "aten::baddbmm")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.key:14s} n={e.count:3d} gpu {e.device_time_total/1e3:7.3f} ms  cpu {e.cpu_time_total/1e3:7.3f} ms  {e.input_shapes}")
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=22, max_name_column_width=50))
nodes = [e for e in prof.key_averages() if e.key.endswith("Backward") or e.key.endswith("Backward0") or e.key.endswith("Backward1") or "Function" in e.key]
nodes.sort(key=lambda e: -e.cpu_time_total)
print("--- autograd nodes by total CPU time")
for e in nodes[:28]:
    print(f"{e.key:44s} n={e.count:4d} cpu {e.cpu_time_total/1e3:7.3f} ms  gpu {e.device_time_total/1e3:7.3f} ms")
# framework-side kernels of the step: which line of the package issues them
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof2:
    step(); torch.cuda.synchronize()
import collections
acc = collections.defaultdict(lambda: [0.0, 0])
for ev in prof2.events():
    if not ev.key.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    where = next((f for f in ev.stack if "anim-nerf_amd" in f or "anim_nerf_amd" in f), "?") if ev.stack else "?"
    k = (ev.key, str(ev.input_shapes)[:60], where.split("anim-nerf_amd/")[-1][:60])
    acc[k][0] += ev.self_device_time_total
    acc[k][1] += 1
print("--- aten ops with device time, by issuing line")
tot = 0.0
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:60]:
    tot += t
    print(f"{t:8.1f} us {c:3d}x  {k[0]:28s} {k[1]:60s} {k[2]}")
print("sum of all aten device time", sum(v[0] for v in acc.values()))
