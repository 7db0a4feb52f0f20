import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import anim_nerf_amd as ana
from anim_nerf_amd import ops, synthetic as syn
from helpers import oracle_table, seeded_model, net_params, golden
from oracle import animnerf_oracle as orc
import test_gpu_training as T
dev = torch.device("cuda:0")
smpl = syn.make_smpl_table(0)
g = golden("render_cfg3_warp_gain")
gain = 50.0
m = seeded_model(smpl, g["seed"], True, gain, g["shift"] * gain / float(g["gain"]), device=dev, mlp_mode="f32")
for p in m.parameters(): p.requires_grad_(False)
n_fine = 8
vr = ana.VolumeRenderer(n_coarse=16, n_fine=n_fine)
pose_np = syn.animated_pose_params(seed=3, bs=2)
names = ("betas", "global_orient", "body_pose", "transl")
pose = {k: torch.from_numpy(pose_np[k]) for k in names}
c2w, focal, cen = syn.pinhole_camera(8, 8)
rays = orc.make_rays(torch.from_numpy(c2w), 8, 8, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(2, 1, 1, 1).view(2, 64, 8)
gen = torch.Generator().manual_seed(4)
t_rgb = torch.rand(2, 64, 3, generator=gen)
z_fine = T._hip_fine_samples(m, vr, rays, pose, dev)
tbl64 = T._fp64(oracle_table(smpl))
st = orc.frame_state(tbl64, T._fp64(pose), T._fp64(T._templ()))
st, rb64 = orc.to_root_frame(st, rays.double())
o2c64 = orc.observation_to_canonical(st)
P = [T._fp64(net_params(n)) for n in (m.nerf, m.nerf_fine)]
for key in ("rgbs_fine", "alphas_fine", "depths_fine", "rgbs", "depths"):
    with torch.no_grad():
        m.set_body_model({k: v.to(dev) for k, v in pose.items()}, T._templ(dev))
        rays_b = m.convert_to_body_model_space(rays.to(dev))
        m.clac_ober2cano_transform()
    rays_b = rays_b.detach().clone().requires_grad_(True)
    m.ober2cano_transform = m.ober2cano_transform.detach().clone().requires_grad_(True)
    res = vr(m, rays_b, perturb=0.0)
    tgt = t_rgb if "rgb" in key else t_rgb[..., :1]
    loss = ((res[key].view(2, 64, -1) - tgt.to(dev)) ** 2).mean()
    loss.backward()
    rb = rays_b.detach().cpu().double().requires_grad_(True)
    st2 = dict(st); st2["ober2cano"] = m.ober2cano_transform.detach().cpu().double().requires_grad_(True)
    field = lambda xyz, fine: orc.field_query(P[1 if fine else 0], xyz, st2, tbl64["lbs_weights"], True, 0.2, chunk=512)
    out = orc.render_rays(field, rb, 16, n_fine, z_fine=z_fine)
    ref = ((out[key] - tgt.double()) ** 2).mean()
    ref.backward()
    a, b = rays_b.grad.cpu().double(), rb.grad
    cols = {"o": slice(0, 3), "d": slice(3, 6), "near": slice(6, 7), "far": slice(7, 8)}
    print(key, "loss", loss.item(), ref.item(), " ".join(f"d_{n} {((a[..., c] - b[..., c]).norm() / (b[..., c].norm() + 1e-300)).item():.1e} (|{b[..., c].norm().item():.1e}|)" for n, c in cols.items()),
          f"d_o2c {((m.ober2cano_transform.grad.cpu().double() - st2['ober2cano'].grad).norm() / st2['ober2cano'].grad.norm()).item():.1e}")

print("---- chain the per-leaf HIP gradients through the fp64 oracle chain and compare with both end-to-end gradients")
key = "rgbs_fine"
tgt = t_rgb
with torch.no_grad():
    m.set_body_model({k: v.to(dev) for k, v in pose.items()}, T._templ(dev))
    rays_b = m.convert_to_body_model_space(rays.to(dev))
    m.clac_ober2cano_transform()
rays_b = rays_b.detach().clone().requires_grad_(True)
m.ober2cano_transform = m.ober2cano_transform.detach().clone().requires_grad_(True)
res = vr(m, rays_b, perturb=0.0)
((res[key].view(2, 64, -1) - tgt.to(dev)) ** 2).mean().backward()
d_rays, d_o2c = rays_b.grad.cpu().double(), m.ober2cano_transform.grad.cpu().double()
p64 = {k: torch.from_numpy(pose_np[k]).double().requires_grad_(True) for k in names}
st = orc.frame_state(tbl64, p64, T._fp64(T._templ()))
st, rb = orc.to_root_frame(st, rays.double())
o2c = orc.observation_to_canonical(st)
((rb * d_rays).sum() + (o2c * d_o2c).sum()).backward()
chained = {k: p64[k].grad.clone() for k in names}
# end to end, HIP
pose_g = {k: torch.from_numpy(pose_np[k]).to(dev).requires_grad_(True) for k in names}
res = ana.system_forward(vr, m, rays.view(2, 8, 8, 8).to(dev), pose_g, T._templ(dev), perturb=0.0, chunk=64)
((res[key].view(2, 64, -1) - tgt.to(dev)) ** 2).mean().backward()
# end to end, oracle fp64
pose_o = {k: torch.from_numpy(pose_np[k]).double().requires_grad_(True) for k in names}
out = orc.render_frame(tbl64, P[0], P[1], rays.double(), pose_o, T._fp64(T._templ()), n_coarse=16, n_fine=n_fine, use_unpose=True, chunk=64, knn_chunk=512, z_fine=z_fine)
((out[key] - tgt.double()) ** 2).mean().backward()
for k in names:
    e2e_h, e2e_o, ch = pose_g[k].grad.cpu().double(), pose_o[k].grad, chained[k]
    print(k, f"hip e2e vs oracle e2e {((e2e_h - e2e_o).norm() / e2e_o.norm()).item():.1e}   chained vs oracle e2e {((ch - e2e_o).norm() / e2e_o.norm()).item():.1e}   hip e2e vs chained {((e2e_h - ch).norm() / ch.norm()).item():.1e}")

print("---- (1) the oracle in fp32 against itself in fp64, same injected samples; (2) fp64 per-leaf gradients chained vs fp64 end to end")
tbl32 = oracle_table(smpl)
pose_32 = {k: torch.from_numpy(pose_np[k]).clone().requires_grad_(True) for k in names}
P32 = [net_params(n) for n in (m.nerf, m.nerf_fine)]
out = orc.render_frame(tbl32, P32[0], P32[1], rays, pose_32, T._templ(), n_coarse=16, n_fine=n_fine, use_unpose=True, chunk=64, knn_chunk=512, z_fine=z_fine.float())
((out[key] - tgt) ** 2).mean().backward()
# fp64 per-leaf
rb_l = rb64.detach().clone().requires_grad_(True)
st3 = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in st.items()}
st3["ober2cano"] = o2c64.detach().clone().requires_grad_(True)
field = lambda xyz, fine: orc.field_query(P[1 if fine else 0], xyz, st3, tbl64["lbs_weights"], True, 0.2, chunk=512)
out = orc.render_rays(field, rb_l, 16, n_fine, z_fine=z_fine)
((out[key] - tgt.double()) ** 2).mean().backward()
p64b = {k: torch.from_numpy(pose_np[k]).double().requires_grad_(True) for k in names}
stb = orc.frame_state(tbl64, p64b, T._fp64(T._templ()))
stb, rbb = orc.to_root_frame(stb, rays.double())
o2cb = orc.observation_to_canonical(stb)
((rbb * rb_l.grad).sum() + (o2cb * st3["ober2cano"].grad).sum()).backward()
for k in names:
    e2e_o = pose_o[k].grad
    print(k, f"oracle fp32 vs fp64 {((pose_32[k].grad.double() - e2e_o).norm() / e2e_o.norm()).item():.1e}   fp64 per-leaf chained vs fp64 e2e {((p64b[k].grad - e2e_o).norm() / e2e_o.norm()).item():.1e}")
