#!/usr/bin/env python3
"""torch.profiler of the per-frame chain alone (SMPL -> root frame -> ober2cano), forward + backward, pose refinement on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True, use_fine=True).to(dev)
F = 16
P = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in syn.animated_pose_params(seed=200, bs=F).items()}
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
c2w, focal, cen = syn.pinhole_camera(32, 32)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 32, 32, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(F, 1, 1, 1).view(F, 1024, 8)
def chain():
    model.set_body_model(P, templ)
    r = model.convert_to_body_model_space(rays)
    model.clac_ober2cano_transform()
    (r.sum() + model.ober2cano_transform.sum()).backward()
for _ in range(3): chain()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): chain()
torch.cuda.synchronize(); print("chain fwd+bwd %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    chain(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=32, max_name_column_width=46))
