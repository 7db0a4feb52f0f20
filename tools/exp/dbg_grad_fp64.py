"""measure: HIP fp32 gradients against fp64 oracle autograd (to set the gates of the tests)"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import anim_nerf_amd as ana
from anim_nerf_amd import ops, synthetic as syn
from helpers import oracle_table, seeded_model, net_params, golden
from oracle import animnerf_oracle as orc
dev = torch.device("cuda:0")
smpl = syn.make_smpl_table(0)
tbl32 = oracle_table(smpl)
tbl64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in tbl32.items()}
templ32 = {k: torch.from_numpy(v) for k, v in syn.template_pose_params().items()}
templ64 = {k: v.double() for k, v in templ32.items()}
templ_d = {k: v.to(dev) for k, v in templ32.items()}
names = ("betas", "global_orient", "body_pose", "transl")
GROUPS = ((0, 10, "betas"), (10, 13, "global_orient"), (13, 82, "body_pose"), (82, 85, "transl"))

# ---- A: the frame chain kernel alone
m = seeded_model(smpl, 3, True, device=dev)
bs, R = 4, 300
pose_np = syn.animated_pose_params(seed=6, bs=bs)
pose = {k: torch.from_numpy(v).to(dev) for k, v in pose_np.items()}
with torch.no_grad():
    m.set_body_model(pose, templ_d)
c = m._chain_consts()
gen = torch.Generator().manual_seed(2)
d_o2c = torch.randn(bs, 6890, 4, 4, generator=gen)
d_rays = torch.randn(bs, R, 8, generator=gen)
rays_w = torch.randn(bs, R, 8, generator=gen)
rays_w[..., 6], rays_w[..., 7] = 0.1 + 3 * torch.rand(bs, R, generator=gen), 3.5 + 3 * torch.rand(bs, R, generator=gen)
args = (pose["betas"].expand(bs, -1).contiguous(), torch.cat([pose["global_orient"], pose["body_pose"]], 1).contiguous(),
        pose["transl"].expand(bs, -1).contiguous(), c["J0"], c["JS"], c["parents"], c["lbs_weights"], c["shapedirs"], c["posedirs"], c["T_template"])
for dt, tb, tp in ((torch.float64, tbl64, templ64), (torch.float32, tbl32, templ32)):
    p = {k: torch.from_numpy(pose_np[k]).to(dt).expand(bs, -1).clone().requires_grad_(True) for k in names}
    st = orc.frame_state(tb, p, tp)
    st, rb = orc.to_root_frame(st, rays_w.to(dt))
    o2c = orc.observation_to_canonical(st)
    for what, L in (("o2c", (o2c * d_o2c.to(dt)).sum()), ("rays", (rb * d_rays.to(dt)).sum())):
        gr = torch.autograd.grad(L, [p[k] for k in names], retain_graph=True, allow_unused=True)
        ref = torch.cat([g if g is not None else torch.zeros_like(p[k]) for g, k in zip(gr, names)], 1)
        kw = dict(d_o2c=d_o2c.to(dev)) if what == "o2c" else dict(d_rays=d_rays.to(dev), rays_world=rays_w.to(dev))
        for fm in (False, True):
            got = ops.frame_backward(*args, **kw, forward_mode=fm).cpu().double()
            whole = ref.abs().max().item()
            print(f"A {str(dt)[6:]} oracle, {what}, forward_mode={fm}: " + "  ".join(
                f"{nm} rel {((got[:, lo:hi] - ref[:, lo:hi].double()).norm() / max(ref[:, lo:hi].norm().item(), 1e-30)):.2e} (|g| {ref[:, lo:hi].abs().max().item():.1e})"
                for lo, hi, nm in GROUPS))

# ---- B: pose refinement through a coarse-only render
for gain in (1.0, 50.0, 3000.0):
    g = golden("render_cfg3_warp_gain")
    m = seeded_model(smpl, g["seed"], True, gain, (0.0, 0.0) if gain == 1.0 else g["shift"] * gain / 3000.0, device=dev, mlp_mode="f32")
    for p_ in m.parameters():
        p_.requires_grad_(False)
    pose_np = syn.animated_pose_params(seed=3, bs=2)
    c2w, focal, cen = syn.pinhole_camera(8, 8)
    rays = orc.make_rays(torch.from_numpy(c2w), 8, 8, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(2, 1, 1, 1)
    gen = torch.Generator().manual_seed(4)
    t_rgb, t_a, t_d = torch.rand(2, 64, 3, generator=gen), (torch.rand(2, 64, 1, generator=gen) > 0.5).float(), 3 + torch.rand(2, 64, 1, generator=gen)
    for n_fine in (0, 8):
        vr = ana.VolumeRenderer(n_coarse=16, n_fine=n_fine)
        sfx = "_fine" if n_fine else ""
        losses = {"rgb": lambda r, dt: ((r["rgbs" + sfx].view(2, 64, 3) - t_rgb.to(r["rgbs"])) ** 2).mean(),
                  "alpha": lambda r, dt: (r["alphas" + sfx].view(2, 64, 1) - t_a.to(r["rgbs"])).abs().mean(),
                  "depth": lambda r, dt: ((r["depths" + sfx].view(2, 64, 1) - t_d.to(r["rgbs"])) ** 2).mean()}
        for lname, lf in losses.items():
            pose_g = {k: torch.from_numpy(pose_np[k]).to(dev).requires_grad_(True) for k in names}
            res = ana.system_forward(vr, m, rays.to(dev), pose_g, templ_d, perturb=0.0, chunk=64)
            lf(res, None).backward()
            line = f"B gain {gain} n_fine {n_fine} loss {lname}: "
            for dt, tb, tp in ((torch.float64, tbl64, templ64), (torch.float32, tbl32, templ32)):
                pose_o = {k: torch.from_numpy(pose_np[k]).to(dt).requires_grad_(True) for k in names}
                P = [{k: v.to(dt) for k, v in net_params(n).items()} for n in (m.nerf, m.nerf_fine)]
                out = orc.render_frame(tb, P[0], P[1], rays.view(2, 64, 8).to(dt), pose_o, tp, n_coarse=16, n_fine=n_fine, use_unpose=True, chunk=64, knn_chunk=512)
                lf(out, dt).backward()
                line += f"[{str(dt)[6:]}] " + " ".join(f"{k} {((pose_g[k].grad.cpu().double() - pose_o[k].grad.double()).norm() / pose_o[k].grad.double().norm()).item():.1e}" for k in names) + "  "
            print(line)
