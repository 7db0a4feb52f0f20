#!/usr/bin/env python3
"""BASELINE configs[0] (plumbing, no GPU): one 200x200 frame, 32 coarse samples, warp on, through the CPU oracle
(oracle/animnerf_oracle.py = the reference's algorithm restated on torch CPU ops).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
from oracle import animnerf_oracle as orc

cores = len(os.sched_getaffinity(0))
torch.set_num_threads(cores)
tbl = syn.make_smpl_table(0)
bm = ana.SMPL(data_struct=tbl)
otbl = dict(v_template=bm.v_template, shapedirs=bm.shapedirs, posedirs=bm.posedirs, J_regressor=bm.J_regressor,
            parents=bm.parents, lbs_weights=bm.lbs_weights, extra_joints_idxs=bm.vertex_joint_selector.extra_joints_idxs)
torch.manual_seed(0)
nc, nf = ana.NeRF(freqs_dir=0, use_view=False), ana.NeRF(freqs_dir=0, use_view=False)
Pc = {k: v.detach() for k, v in nc.named_parameters()}
Pf = {k: v.detach() for k, v in nf.named_parameters()}
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 200
c2w, focal, cen = syn.pinhole_camera(H, W)
rays = orc.make_rays(torch.from_numpy(c2w), H, W, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
pose = {k: torch.from_numpy(v) for k, v in syn.animated_pose_params(seed=100).items()}
templ = {k: torch.from_numpy(v) for k, v in syn.template_pose_params().items()}
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000           # rays timed (every k-th ray of the frame)
stride = max(1, rays.shape[1] // n)
sample = rays[:, ::stride][:, :n].contiguous()
kw = dict(n_coarse=32, n_fine=0, use_unpose=True, chunk=256, knn_chunk=2048)
orc.render_frame(otbl, Pc, Pf, sample[:, :64], pose, templ, **kw)
t0 = time.perf_counter()
out = orc.render_frame(otbl, Pc, Pf, sample, pose, templ, **kw)
dt = time.perf_counter() - t0
print(json.dumps({"workload": f"BASELINE configs[0]: {H}x{W}, 32 coarse, warp on, CPU oracle", "rays_timed": sample.shape[1],
                  "seconds": dt, "rays_per_s": sample.shape[1] / dt, "cores": cores,
                  "frame_seconds_extrapolated": H * W / (sample.shape[1] / dt), "alpha_max": out["alphas"].max().item()}))
if torch.cuda.is_available():                                  # the same frame through the HIP path, and the comparison
    dev = torch.device("cuda:0")
    m = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True, use_fine=True, mlp_mode="f32").eval().to(dev)
    m.nerf.load_state_dict(nc.state_dict()); m.nerf_fine.load_state_dict(nf.state_dict())
    vr = ana.VolumeRenderer(n_coarse=32, n_fine=0)
    to = lambda d: {k: v.to(dev) for k, v in d.items()}
    full = rays.to(dev)
    ana.batched_inference(vr, m, full, to(pose), to(templ), chunk=1 << 20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        got = ana.batched_inference(vr, m, full, to(pose), to(templ), chunk=1 << 20)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    sub = got["rgbs"][:, ::stride][:, :n].cpu()
    err = (sub - out["rgbs"]).abs().max(-1).values / out["rgbs"].abs().max(-1).values.clamp_min(1e-3)
    print(json.dumps({"workload": f"same frame, HIP path (fp32 parity mode), whole {H}x{W} frame", "seconds_per_frame": dt,
                      "rays_per_s": H * W / dt, "rays_within_1e-4_of_oracle": (err <= 1e-4).float().mean().item(),
                      "max_rel_err": err.max().item()}))
