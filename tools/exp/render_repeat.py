#!/usr/bin/env python3
"""Render ONE 1024 x 1024 frame with the animated-SMPL warp (configs[2]'s shape, bf16) N times next to a busy side stream and
compare every output with the first render's, bit for bit.  The renderer has no float atomics: any difference is a fault.
    python tools/exp/render_repeat.py [N=40] [hw=1024]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
torch.manual_seed(0)
m = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True, use_fine=True, mlp_mode="bf16").to(dev)
with torch.no_grad():                                    # some density on the body: scale the sigma rows
    for net in (m.nerf, m.nerf_fine):
        net.sigma.weight.mul_(300.0); net.sigma.bias.add_(2.0)
pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=57, bs=1, pose_std=0.35, transl_z=-2.6).items()}
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
c2w, focal, cen = syn.pinhole_camera(hw, hw)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
vr = ana.VolumeRenderer(n_coarse=64, n_fine=64)
side = torch.cuda.Stream()
a = torch.randn(2048, 2048, device=dev)
first, bad = None, 0
t0 = time.perf_counter()
for it in range(N):
    with torch.cuda.stream(side):
        for _ in range(6):
            a = torch.tanh(a @ a) * 0.5
    with torch.no_grad():
        out = ana.batched_inference(vr, m, rays, pose, templ, chunk=1 << 20)
    if first is None:
        first = {k: v.clone() for k, v in out.items()}
        print("covered pixels", float((first["alphas_fine"] > 0.5).float().mean()), flush=True)
        continue
    for k in first:
        if not torch.equal(out[k], first[k]):
            bad += 1
            d = (out[k].float() - first[k].float()).abs()
            print(f"render {it}: {k} differs in {int((d > 0).sum())} entries, max {float(d.max()):.3e}", flush=True)
torch.cuda.synchronize()
print(f"{N} renders of one {hw} x {hw} frame next to a busy stream: {bad} outputs differed from the first render's ; {time.perf_counter() - t0:.1f} s")
