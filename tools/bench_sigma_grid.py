#!/usr/bin/env python3
"""BASELINE configs[4] (mesh extraction input): relu(sigma) of the fine field on an N^3 grid around the posed body
(extract_mesh.py:152-158) through `sigma_grid`.  Sigma rows are rescaled about their median like in bench.py so that the
grid has occupied voxels.  Usage: python tools/bench_sigma_grid.py [N ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn

dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
torch.manual_seed(0)
model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True,
                     use_fine=True, mlp_mode="bf16").eval().to(dev)
g = torch.Generator().manual_seed(5)
probe = (torch.rand(1, 4096, 3, generator=g) * 1.6 - 0.8).to(dev)
with torch.no_grad():
    for net in (model.nerf, model.nerf_fine):
        net.mlp_mode = "f32"; med = net(probe)[1].median().item(); net.mlp_mode = "bf16"
        net.sigma.weight.mul_(3000.0); net.sigma.bias.mul_(3000.0).add_(-3000.0 * med)
pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=100).items()}
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
rays = torch.zeros(1, 1, 8, device=dev); rays[..., 5] = -1; rays[..., 7] = 10
with torch.no_grad():
    model.set_body_model(pose, templ)
    model.convert_to_body_model_space(rays)
    model.clac_ober2cano_transform()
    for N in [int(a) for a in sys.argv[1:]] or [256, 512]:
        for dense in (False, True):
            model.skip_invalid_samples = not dense
            ana.sigma_grid(model, 64)                                   # warm-up
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sig, _ = ana.sigma_grid(model, N, chunk=1 << 25)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(json.dumps({"workload": f"BASELINE configs[4]: {N}^3 sigma grid, bf16, 1 GPU", "mlp_on_valid_voxels_only": not dense,
                              "seconds": dt, "points_per_s": N ** 3 / dt, "occupied_voxels": int((sig > 0).sum()),
                              "threshold_20_voxels": int((sig > 20).sum())}))
