import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
from helpers import golden, seeded_model, net_params, oracle_table
from oracle import animnerf_oracle as orc
from test_gpu_training import _hip_fine_samples, _templ, _fp64
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
g = golden("render_cfg3_warp_gain")
gain = 50.0
m = seeded_model(tbl, g["seed"], True, gain, g["shift"] * gain / float(g["gain"]), device=dev, mlp_mode="f32")
for p in m.parameters(): p.requires_grad_(False)
vr = ana.VolumeRenderer(n_coarse=16, n_fine=8)
pose = {k: torch.from_numpy(v) for k, v in syn.animated_pose_params(seed=3, bs=2).items()}
c2w, focal, cen = syn.pinhole_camera(8, 8)
rays = orc.make_rays(torch.from_numpy(c2w), 8, 8, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(2, 1, 1, 1)
gen = torch.Generator().manual_seed(4)
target = torch.rand(2, 64, 3, generator=gen)
tbl64 = _fp64(oracle_table(tbl))
P = [_fp64(net_params(n)) for n in (m.nerf, m.nerf_fine)]
loss_of = lambda res: ((res["rgbs_fine"].reshape(target.shape) - target.to(res["rgbs_fine"])) ** 2).mean()
G = {}
for off in ("", "1"):
    if off: os.environ["ANR_FRAME_SETUP_OFF"] = "1"
    else: os.environ.pop("ANR_FRAME_SETUP_OFF", None)
    zf = _hip_fine_samples(m, vr, rays, pose, dev)
    with torch.no_grad():
        rb = m.frame_setup({k: v.to(dev) for k, v in pose.items()}, _templ(dev), rays.view(2, 64, 8).to(dev))
    rbg = rb.detach().clone().requires_grad_(True)
    m.ober2cano_transform = m.ober2cano_transform.detach().clone().requires_grad_(True)
    loss = loss_of(vr(m, rbg, perturb=0.0)); loss.backward()
    st = dict(verts=m.verts.detach().cpu().double(), ober2cano=m.ober2cano_transform.detach().cpu().double().requires_grad_(True))
    rbo = rbg.detach().cpu().double().requires_grad_(True)
    field = lambda xyz, fine: orc.field_query(P[1 if fine else 0], xyz, st, tbl64["lbs_weights"], True, 0.2, chunk=512)
    ref = loss_of(orc.render_rays(field, rbo, 16, 8, z_fine=zf)); ref.backward()
    G[off] = (rbg.grad.cpu().double(), rbo.grad, m.ober2cano_transform.grad.cpu().double(), st["ober2cano"].grad)
    hip, orc_g = G[off][0], G[off][1]
    per_ray = (hip[..., :3] - orc_g[..., :3]).norm(dim=-1) / orc_g[..., :3].norm()
    worst = torch.topk(per_ray.flatten(), 4)
    print("setup off" if off else "fused", "loss", loss.item(), ref.item(), "rel o'", ((hip[..., :3] - orc_g[..., :3]).norm() / orc_g[..., :3].norm()).item(),
          "worst rays", worst.indices.tolist(), [f"{v:.1e}" for v in worst.values.tolist()], "o2c rel", ((G[off][2] - G[off][3]).norm() / G[off][3].norm()).item())
print("HIP d o' fused vs separate:", ((G[""][0][..., :3] - G["1"][0][..., :3]).norm() / G["1"][0][..., :3].norm()).item(),
      "oracle:", ((G[""][1][..., :3] - G["1"][1][..., :3]).norm() / G["1"][1][..., :3].norm()).item())
# ---- ray 116 (frame 1, ray 52) under the fused setup: sample by sample
os.environ.pop("ANR_FRAME_SETUP_OFF", None)
from anim_nerf_amd import ops
zf = _hip_fine_samples(m, vr, rays, pose, dev)
with torch.no_grad():
    rb = m.frame_setup({k: v.to(dev) for k, v in pose.items()}, _templ(dev), rays.view(2, 64, 8).to(dev))
    zc = vr.sample_coarse(rb)
    zs = torch.sort(torch.cat([zc, zf.float().to(dev)], -1), -1).values
    for name, z_, net, Pn in (("coarse", zc, m.nerf, P[0]), ("fine", zs, m.nerf_fine, P[1])):
        b, r = 1, 52
        zz = z_[b:b + 1, r:r + 1]
        rr = rb[b:b + 1, r:r + 1]
        xyz = (rr[..., None, :3] + zz[..., None] * rr[..., None, 3:6]).reshape(1, -1, 3)
        pts, dist_h, idx_h, bl_h = ops.warp_points(m.knn_index(), m.ober2cano_transform.detach()[b:b + 1], m.body_model.lbs_weights, 0.2, xyz=xyz.contiguous(), debug=True) if False else (None, None, None, None)
        hip4 = m.warped_points(rays=rb, z=z_).view(2, 64, -1, 4)[b, r].cpu().double()
        xc, valid_o, dbg = orc.warp_to_canonical(xyz.cpu().double(), m.verts.cpu().double()[b:b + 1], tbl64["lbs_weights"], m.ober2cano_transform.detach().cpu().double()[b:b + 1], 0.2, chunk=512)
        sig32 = net.eval_points(hip4.float().to(dev), "f32")[:, 3].cpu().double()
        sig64 = orc.mlp_forward(Pn, hip4[None, :, :3])[1].view(-1)
        from anim_nerf_amd import ops as _o
        _p, nidx, nw = _o.warp_points(m.knn_index(), m.ober2cano_transform.detach(), m.body_model.lbs_weights, 0.2, rays=rb, z=z_, skip_far=True, neighbours=True)
        K_ = z_.shape[-1]
        nidx, nw = nidx.view(2, 64, K_, 4)[b, r].cpu(), nw.view(2, 64, K_, 4)[b, r].cpu()
        db_, Tb_, w_o = orc.blend_neighbours(dbg["dist"], dbg["idx"], tbl64["lbs_weights"], m.ober2cano_transform.detach().cpu().double()[b:b + 1])
        for si in range(K_):
            if hip4[si, 3] > 0:
                print(name, si, "hip idx", nidx[si].tolist(), "w", [round(v, 5) for v in nw[si].tolist()], "| orc idx", dbg["idx"][0, si].tolist(), "w", [round(v, 5) for v in (w_o[0, si] if w_o is not None else torch.zeros(4)).view(-1).tolist()])
        print(name, "valid hip", hip4[:, 3].int().tolist())
        print(name, "valid orc", valid_o.view(-1).int().tolist())
        print(name, "blended dist", [round(v, 5) for v in dbg["blended"].view(-1).tolist()])
        print(name, "d0,d1", [(round(a, 6), round(b_, 6)) for a, b_ in dbg["dist"][0, :, :2].tolist()])
        print(name, "|xc diff|", [f"{v:.1e}" for v in (hip4[:, :3] - xc[0]).abs().max(-1).values.tolist()])
        print(name, "sigma32", [round(v, 4) for v in sig32.tolist()])
        print(name, "sigma64", [round(v, 4) for v in sig64.tolist()])
