import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import anim_nerf_amd as ana
from anim_nerf_amd import ops, synthetic as syn
from helpers import oracle_table, seeded_model, net_params, golden
from oracle import animnerf_oracle as orc
import test_gpu_training as T
dev = torch.device("cuda:0")
smpl = syn.make_smpl_table(0)
g = golden("render_cfg3_warp_gain")
gain = 50.0
m = seeded_model(smpl, g["seed"], True, gain, g["shift"] * gain / float(g["gain"]), device=dev, mlp_mode="f32")
for p in m.parameters(): p.requires_grad_(False)
n_fine = 8
vr = ana.VolumeRenderer(n_coarse=16, n_fine=n_fine)
pose_np = syn.animated_pose_params(seed=3, bs=2)
c2w, focal, cen = syn.pinhole_camera(8, 8)
rays = orc.make_rays(torch.from_numpy(c2w), 8, 8, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(2, 1, 1, 1)
names = ("betas", "global_orient", "body_pose", "transl")
gen = torch.Generator().manual_seed(4)
t_rgb = torch.rand(2, 64, 3, generator=gen)
for key in ("rgbs_fine", "alphas_fine", "depths_fine", "rgbs"):
    pose_g = {k: torch.from_numpy(pose_np[k]).to(dev).requires_grad_(True) for k in names}
    res = ana.system_forward(vr, m, rays.to(dev), pose_g, T._templ(dev), perturb=0.0, chunk=64)
    tgt = t_rgb if "rgb" in key else t_rgb[..., :1]
    loss = ((res[key].view(2, 64, -1) - tgt.to(dev)) ** 2).mean()
    loss.backward()
    z_fine = T._hip_fine_samples(m, vr, rays, pose_g, dev)
    for inj in (True, False):
        pose_o = {k: torch.from_numpy(pose_np[k]).double().requires_grad_(True) for k in names}
        P = [T._fp64(net_params(n)) for n in (m.nerf, m.nerf_fine)]
        out = orc.render_frame(T._fp64(oracle_table(smpl)), P[0], P[1], rays.view(2, 64, 8).double(), pose_o, T._fp64(T._templ()),
                               n_coarse=16, n_fine=n_fine, use_unpose=True, chunk=64, knn_chunk=512, z_fine=z_fine if inj else None)
        ref = ((out[key] - tgt.double()) ** 2).mean()
        ref.backward()
        dz = (out["_z_fine"] - z_fine).abs()
        print(key, "inject" if inj else "own   ", "loss", loss.item(), ref.item(), "z_fine differs by > 1e-5 on", int((dz > 1e-5).sum()), "of", dz.numel(),
              " ".join(f"{k} {((pose_g[k].grad.cpu().double() - pose_o[k].grad).norm() / pose_o[k].grad.norm()).item():.1e}" for k in names))
