#!/usr/bin/env python3
"""torch.profiler table of one training step (where the backward's library GEMMs and glue go)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
torch.manual_seed(0)
model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_fine=True, mlp_mode=mode).to(dev)
tr = ana.Trainer(model, ana.VolumeRenderer(n_coarse=64, n_fine=32), ana.TrainHParams())
F = 16
c2w, focal, cen = syn.pinhole_camera(32, 32)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 32, 32, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(F, 1, 1, 1)
pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=2, bs=F).items()}
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
rgbs = torch.rand(F, 32, 32, 3, device=dev); alphas = (torch.rand(F, 32, 32, 1, device=dev) > 0.5).float()
fg = torch.rand(F, 128, 3, device=dev) * 0.4 - 0.2; bg = torch.rand(F, 128, 3, device=dev) * 2 - 1
for _ in range(2):
    tr.step(rays, rgbs, alphas, pose, templ, fg, bg)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(rays, rgbs, alphas, pose, templ, fg, bg)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by=(sys.argv[2] if len(sys.argv) > 2 else "cuda_time_total"), row_limit=30, max_name_column_width=60))
try:
    a = torch.randn(1000, 256, device=dev, dtype=torch.bfloat16)
    print("mm out_dtype:", torch.mm(a.t(), a, out_dtype=torch.float32).dtype)
except Exception as e:
    print("mm out_dtype unsupported:", e)
