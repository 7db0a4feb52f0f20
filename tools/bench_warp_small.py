#!/usr/bin/env python3
"""The warp on a training-shaped batch (BASELINE configs[3]: 16 bodies x 1,024 random pixels x 64 samples): time per call of
the small-batch path, eight lanes per sample (warp_search_groups_kernel) against a lane per sample (ANR_WARP_LANE_PER_SAMPLE=1).
    python tools/bench_warp_small.py [reps] [groups|lanes|both] [bodies]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import anim_nerf_amd as ana                                  # noqa: E402
from anim_nerf_amd import synthetic as syn                   # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else "both"
dev = torch.device("cuda:0")
bs, n_rays, K, hw = (int(sys.argv[3]) if len(sys.argv) > 3 else 16), 1024, 64, 32
tbl = syn.make_smpl_table(0)
torch.manual_seed(3)
m = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_fine=True).eval().to(dev)
pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=200, bs=bs, pose_std=0.3).items()}
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
c2w, focal, cen = syn.pinhole_camera(hw, hw)
full = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(-1, 8)
pick = torch.arange(hw * hw, device=dev)[None].repeat(bs, 1)          # bench.py's cfg4: the 32 x 32 image of every frame
with torch.no_grad():
    m.set_body_model(pose, templ)
    rays = m.convert_to_body_model_space(full[pick].contiguous())
    m.clac_ober2cano_transform()
    z = ana.VolumeRenderer(n_coarse=K).sample_coarse(rays)
    if os.environ.get("WARP_BENCH_FAR"):                      # every sample far from the body: the call's fixed cost
        z = z + 100.0
    args = (m.knn_index(), m.ober2cano_transform, m.body_model.lbs_weights, 0.2)
    for name in (("groups", "lanes") if which == "both" else (which,)):
        if name == "lanes":
            os.environ["ANR_WARP_LANE_PER_SAMPLE"] = "1"
        else:
            os.environ.pop("ANR_WARP_LANE_PER_SAMPLE", None)
        out = ana.ops.warp_points(*args, rays=rays, z=z, skip_far=True, lean=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = ana.ops.warp_points(*args, rays=rays, z=z, skip_far=True, lean=True)
        e1.record()
        torch.cuda.synchronize()
        xyz = ana.ops.points_from_rays(rays, z)
        lo, hi = m.verts.amin(1, keepdim=True) - 0.2, m.verts.amax(1, keepdim=True) + 0.2
        near = ((xyz[..., :3] >= lo) & (xyz[..., :3] <= hi)).all(-1)
        print(f"{name:7s} {e0.elapsed_time(e1) / reps:7.3f} ms per call   samples {bs * n_rays * K}  in the padded box "
              f"{int(near.sum())}  valid {int(out[3].item())}")
