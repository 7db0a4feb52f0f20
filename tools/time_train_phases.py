import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
from anim_nerf_amd.render import system_forward
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
torch.manual_seed(0)
model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_fine=True, mlp_mode="bf16").to(dev)
hp = ana.TrainHParams()
tr = ana.Trainer(model, ana.VolumeRenderer(n_coarse=64, n_fine=32), hp)
F = 16
c2w, focal, cen = syn.pinhole_camera(32, 32)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 32, 32, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(F, 1, 1, 1)
pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=2, bs=F).items()}
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
rgbs = torch.rand(F, 32, 32, 3, device=dev); alphas = (torch.rand(F, 32, 32, 1, device=dev) > 0.5).float()
fg = torch.rand(F, 128, 3, device=dev) * 0.4 - 0.2; bg = torch.rand(F, 128, 3, device=dev) * 2 - 1
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(4):
    t0 = sync()
    tr.optimizer.zero_grad(set_to_none=True)
    model.set_body_model(pose, templ); t1 = sync()
    flat = model.convert_to_body_model_space(rays.view(F, 1024, 8)); model.clac_ober2cano_transform(); model.knn_index(); t2 = sync()
    res = ana.render_prepared(tr.renderer, model, flat, chunk=2048, perturb=1.0); t3 = sync()
    res = {k: v.view(F, 32, 32, -1) for k, v in res.items()}
    loss, det = ana.compute_loss(model, hp, rgbs, alphas, res, fg, bg); t4 = sync()
    loss.backward(); t5 = sync()
    tr.optimizer.step(); t6 = sync()
    print(f"smpl {1e3*(t1-t0):6.1f}  frame-prep {1e3*(t2-t1):6.1f}  render-fwd {1e3*(t3-t2):6.1f}  loss {1e3*(t4-t3):6.1f}  backward {1e3*(t5-t4):6.1f}  adam {1e3*(t6-t5):6.1f}  total {1e3*(t6-t0):6.1f} ms")
