import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
dev = torch.device('cuda:0')
tbl = syn.make_smpl_table(0)
torch.manual_seed(3)
m = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_fine=False).to(dev)
with torch.no_grad():
    m.nerf.sigma.weight.mul_(300.0)
hp = ana.TrainHParams(n_samples=16, n_importance=0, chunk=512, lr=1e-3)
tr = ana.Trainer(m, ana.VolumeRenderer(n_coarse=16, n_fine=0), hp)
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
c2w, focal, cen = syn.pinhole_camera(8, 8)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 8, 8, focal.tolist(), 0.1, 10.0, cen.tolist())[None]
pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=2).items()}
gen = torch.Generator().manual_seed(0)
tgt, alp = torch.rand(1, 8, 8, 3, generator=gen).to(dev), torch.ones(1, 8, 8, 1, device=dev)
fg = (torch.rand(1, 64, 3, generator=gen) * 0.4 - 0.2).to(dev)
bg = (torch.rand(1, 64, 3, generator=gen) * 2 - 1).to(dev)
w0 = [p.detach().clone() for p in tr.params]
loss, details = tr.step(rays, tgt, alp, pose, templ, fg, bg, perturb=0.0)
print('loss', loss.item(), {k: float(v) for k, v in details.items()})
print('grad norms', [float(p.grad.norm()) if p.grad is not None else None for p in tr.params][:6])
print('delta', [float((p.detach() - w).abs().max()) for p, w in zip(tr.params, w0)][:6])
pts = torch.cat([torch.rand(256, 3, device=dev) * 2 - 1, torch.ones(256, 1, device=dev)], -1)
with torch.no_grad():
    a = m.nerf.eval_points(pts).clone()
w1 = [p.detach().clone() for p in tr.params]
loss, details = tr.step(rays, tgt, alp, pose, templ, fg, bg, perturb=0.0)
print('delta2', [float((p.detach() - w).abs().max()) for p, w in zip(tr.params, w1)][:6])
with torch.no_grad():
    b = m.nerf.eval_points(pts).clone()
    m.nerf._pack_cache.clear()
    c = m.nerf.eval_points(pts).clone()
print('a==b', torch.equal(a, b), 'b==c', torch.equal(b, c), (a - c).abs().max().item())
print(a[:3], c[:3])
for mode in ('f32',):
    with torch.no_grad():
        print(mode, m.nerf.eval_points(pts, mode)[:3])
