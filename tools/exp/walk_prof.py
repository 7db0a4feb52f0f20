#!/usr/bin/env python3
"""What does a wavefront of the lane-per-sample neighbour search (warp_search_kernel, the cfg3 / cfg5 path) execute per item?
Needs the experiment build
    python anim-nerf_amd/build.py -DANR_SEARCH_PROF --out=anim-nerf_amd/libanimnerf_hip.prof.so
    ANIMNERF_HIP_LIB=$PWD/anim-nerf_amd/libanimnerf_hip.prof.so python tools/exp/walk_prof.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn

dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
torch.manual_seed(0)
model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True, use_fine=True, mlp_mode="bf16").eval().to(dev)
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
hw = 1024
c2w, focal, cen = syn.pinhole_camera(hw, hw)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=100).items()}
vr = ana.VolumeRenderer(n_coarse=64, n_fine=64)
lib = ana._lib.load()
buf = (ctypes.c_ulonglong * 8)()
lib.anr_walk_prof_read.argtypes = [ctypes.c_void_p]
names = ["wavefront items", "top box tests", "super box tests", "cluster box tests", "cluster scans", "lanes needing a scan", "seed passes", "active lanes"]
with torch.no_grad():
    model.set_body_model(pose, templ)
    r = model.convert_to_body_model_space(rays)
    model.clac_ober2cano_transform()
    z = vr.sample_coarse(r)
    torch.cuda.synchronize(); lib.anr_walk_prof_read(buf)
    pts = model.warped_points(rays=r, z=z, skip_far=True)
    torch.cuda.synchronize(); lib.anr_walk_prof_read(buf)
    a = np.array(list(buf), dtype=np.float64)
    print(f"coarse pass of a cfg3 frame (2^20 rays x 64 samples), all search kernels of the call (cells + samples):")
    for n, v in zip(names, a):
        print(f"  {n:22s} {v:14.0f}   per item {v / a[0]:8.2f}")
    print(f"  lanes per scan {a[5] / max(a[4], 1):.1f} of 64")
