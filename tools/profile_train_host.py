#!/usr/bin/env python3
"""Host-side profile (cProfile) of the cfg4 training step: where the Python time goes once the GPU work is short."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
tbl = syn.make_smpl_table(0)
torch.manual_seed(0)
model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True, use_fine=True, mlp_mode="bf16").to(dev)
hp = ana.TrainHParams(n_samples=64, n_importance=32, chunk=2048)
F = 16
table = ana.BodyModelParams(114).to(dev)
seeded = syn.animated_pose_params(seed=200, bs=114)
for name in table.param_names:
    table.init_parameters(name, torch.from_numpy(seeded[name]).to(dev), requires_grad=True)
trainer = ana.Trainer(model, ana.VolumeRenderer(n_coarse=64, n_fine=32), hp, body_model_params=table)
frame_idx = torch.arange(F, device=dev) * (114 // F)
c2w, focal, cen = syn.pinhole_camera(32, 32)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 32, 32, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(F, 1, 1, 1)
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
g = torch.Generator().manual_seed(0)
rgbs = torch.rand(F, 32, 32, 3, generator=g).to(dev)
alphas = (torch.rand(F, 32, 32, 1, generator=g) > 0.5).float().to(dev)
fg = (torch.rand(F, 128, 3, generator=g) * 0.4 - 0.2).to(dev)
bg = (torch.rand(F, 128, 3, generator=g) * 2 - 1).to(dev)
step = lambda: trainer.step(rays, rgbs, alphas, None, templ, fg, bg, perturb=1.0, frame_idx=frame_idx)
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"10 steps: host returned after {t_host*100:.2f} ms/step, GPU done after {t_all*100:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
