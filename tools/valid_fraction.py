#!/usr/bin/env python3
"""Fraction of ray samples within dis_threshold of the body (valid = 1) for the cfg3 / cfg4 synthetic workloads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import ops, synthetic as syn

dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
torch.manual_seed(0)
model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True,
                     use_fine=True, mlp_mode="bf16").eval().to(dev)
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
for hw, seed in ((512, 100), (32, 200)):
    c2w, focal, cen = syn.pinhole_camera(hw, hw)
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
    pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=seed).items()}
    vr = ana.VolumeRenderer(n_coarse=64, n_fine=64)
    with torch.no_grad():
        model.set_body_model(pose, templ)
        r = model.convert_to_body_model_space(rays)
        model.clac_ober2cano_transform()
        z = vr.sample_coarse(r)
        pts = model.warped_points(rays=r, z=z, skip_far=True)
        v = pts[:, 3].view(-1, 64)
        print(f"{hw}x{hw}: valid samples {v.mean().item():.4f}; rays with any valid {(v.sum(1) > 0).float().mean().item():.4f}; "
              f"32-point tiles with any valid {(pts[:, 3].view(-1, 32).sum(1) > 0).float().mean().item():.4f}")
