#!/usr/bin/env python3
"""Where does a wavefront of warp_search_groups_kernel spend its time?  Needs the experiment build
    python anim-nerf_amd/build.py -DANR_SEARCH_PROF --out=anim-nerf_amd/libanimnerf_hip.prof.so
    ANIMNERF_HIP_LIB=$PWD/anim-nerf_amd/libanimnerf_hip.prof.so python tools/exp/search_prof.py [bodies]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
torch.manual_seed(3)
m = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_fine=True).eval().to(dev)
pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=200, bs=bs, pose_std=0.3).items()}
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
c2w, focal, cen = syn.pinhole_camera(32, 32)
full = ana.gen_rays(torch.from_numpy(c2w).to(dev), 32, 32, focal.tolist(), 0.1, 10.0, cen.tolist()).view(-1, 8)
with torch.no_grad():
    m.set_body_model(pose, templ)
    rays = m.convert_to_body_model_space(full[None].repeat(bs, 1, 1).contiguous())
    m.clac_ober2cano_transform()
    z = ana.VolumeRenderer(n_coarse=64).sample_coarse(rays)
    args = (m.knn_index(), m.ober2cano_transform, m.body_model.lbs_weights, 0.2)
    for _ in range(3):
        ana.ops.warp_points(*args, rays=rays, z=z, skip_far=True, neighbours=True)
    torch.cuda.synchronize()
lib = ana._lib.load()
buf = (ctypes.c_longlong * (8192 * 8))()
lib.anr_search_prof_read.argtypes = [ctypes.c_void_p]
assert lib.anr_search_prof_read(buf) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(8192, 8)[:256 * 4]
names = ["total", "in trips", "atomics", "hand-out iterations", "hop barrier", "-", "steps", "cycles"]
tick = 1e-2                                                   # wall_clock64: 100 MHz -> 10 ns
print(f"bodies {bs}: {len(a)} wavefronts")
for i, nm in enumerate(names):
    v = a[:, i].astype(np.float64) * (tick if i in (0, 1, 4) else 1.0)
    unit = "us" if i in (0, 1, 4) else ""
    print(f"  {nm:9s} min {v.min():9.1f}  median {np.median(v):9.1f}  mean {v.mean():9.1f}  max {v.max():9.1f} {unit}   sum {v.sum():12.0f}")
print("  shader clock over the kernel: %.0f MHz" % (a[:, 7].sum() / (a[:, 0].sum() * tick)))
busy = a[:, 0] > 0
print("  wavefronts that ran:", int(busy.sum()))
order = np.argsort(-a[:, 0])[:5]
print("  slowest wavefronts:", [(int(i), (a[i] * np.array([tick, tick, 1, 1, tick, 1, 1, 1])).round(1).tolist()) for i in order])

ev = (ctypes.c_longlong * (64 * 256))()
lib.anr_search_events_read.argtypes = [ctypes.c_void_p]
assert lib.anr_search_events_read(ev) == 0
e = np.frombuffer(ev, dtype=np.int64).reshape(64, 128, 2)
tags = {1: "start", 2: "staged", 3: "trip>", 4: "<trip", 5: "step", 6: "flush>", 7: "<flush", 8: "bar>", 9: "<bar", 20: "hop0", 21: "hop1"}
for w in (0, 1, 5, 17):
    t0 = e[w, 0, 1]
    line = []
    for tag, t in e[w]:
        if tag == 0:
            break
        line.append(f"{tags.get(int(tag), int(tag))}@{(t - t0) * tick:.1f}")
    print(f"wave {w}: " + " ".join(line))
