import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0); torch.manual_seed(0)
model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_fine=True, mlp_mode="bf16").to(dev)
F = 16
table = ana.BodyModelParams(114).to(dev)
seeded = syn.animated_pose_params(seed=200, bs=114)
for name in table.param_names:
    table.init_parameters(name, torch.from_numpy(seeded[name]).to(dev), requires_grad=True)
tr = ana.Trainer(model, ana.VolumeRenderer(n_coarse=64, n_fine=32), ana.TrainHParams(), table)
c2w, focal, cen = syn.pinhole_camera(32, 32)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 32, 32, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(F, 1, 1, 1)
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
rgbs = torch.rand(F, 32, 32, 3, device=dev); alphas = (torch.rand(F, 32, 32, 1, device=dev) > 0.5).float()
fidx = torch.arange(F, device=dev) * 7
for _ in range(2): tr.step(rays, rgbs, alphas, None, templ, perturb=1.0, frame_idx=fidx)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(rays, rgbs, alphas, None, templ, perturb=1.0, frame_idx=fidx); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=55))
