#!/usr/bin/env python3
"""Sample statistics of the cfg3 synthetic frame: fraction of samples inside the body's bounding box + dis_threshold
(those are searched) and fraction within dis_threshold of the surface (valid: those run through the MLP)."""
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
gain = 3000.0
g = torch.Generator().manual_seed(5)
probe = (torch.rand(1, 4096, 3, generator=g) * 1.6 - 0.8).to(dev)
with torch.no_grad():
    for net in (model.nerf, model.nerf_fine):
        net.mlp_mode = "f32"; med = net(probe)[1].median().item(); net.mlp_mode = "bf16"
        net.sigma.weight.mul_(gain); net.sigma.bias.mul_(gain).add_(-gain * med)
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
hw = 512
c2w, focal, cen = syn.pinhole_camera(hw, hw)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=100).items()}
vr = ana.VolumeRenderer(n_coarse=64, n_fine=64)
with torch.no_grad():
    model.set_body_model(pose, templ)
    r = model.convert_to_body_model_space(rays)
    model.clac_ober2cano_transform()
    lo, hi = model.verts[0].min(0).values, model.verts[0].max(0).values
    z = vr.sample_coarse(r)
    w, _, _, _ = vr._shade(model, r, z, True, 0.0, True)
    z_all = vr.sample_fine_sorted(z, w, 0.0)
    for name, zz in (("coarse", z), ("fine", z_all)):
        K = zz.shape[-1]
        p = r[0, :, None, :3] + zz[0, :, :, None] * r[0, :, None, 3:6]
        d = torch.clamp(torch.maximum(lo - p, p - hi), min=0).pow(2).sum(-1)
        near = d < 0.04
        pts = model.warped_points(rays=r, z=zz, skip_far=True)
        valid = pts[:, 3].view(-1, K) >= 1
        print(f"{name}: K={K} near-box fraction {near.float().mean().item():.4f} valid fraction {valid.float().mean().item():.4f} "
              f"valid/near {valid.float().sum().item() / near.float().sum().item():.3f}")
