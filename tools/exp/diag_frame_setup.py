import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
from helpers import golden, seeded_model
from oracle import animnerf_oracle as orc
from test_gpu_training import _hip_fine_samples, _templ
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
g = golden("render_cfg3_warp_gain")
gain = 50.0
m = seeded_model(tbl, g["seed"], True, gain, g["shift"] * gain / float(g["gain"]), device=dev, mlp_mode="f32")
vr = ana.VolumeRenderer(n_coarse=16, n_fine=8)
pose = {k: torch.from_numpy(v) for k, v in syn.animated_pose_params(seed=3, bs=2).items()}
c2w, focal, cen = syn.pinhole_camera(8, 8)
rays = orc.make_rays(torch.from_numpy(c2w), 8, 8, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(2, 1, 1, 1)
out = {}
for off in ("", "1"):
    if off: os.environ["ANR_FRAME_SETUP_OFF"] = "1"
    else: os.environ.pop("ANR_FRAME_SETUP_OFF", None)
    zf = _hip_fine_samples(m, vr, rays, pose, dev)
    with torch.no_grad():
        rb = m.frame_setup({k: v.to(dev) for k, v in pose.items()}, _templ(dev), rays.view(2, 64, 8).to(dev))
        res = vr(m, rb, perturb=0.0)
    vr.record = {}
    rbg = rb.detach().clone().requires_grad_(True)
    m.ober2cano_transform = m.ober2cano_transform.detach().clone().requires_grad_(True)
    vr(m, rbg, perturb=0.0)
    zs_fwd = vr.record["z_sorted"][0].cpu().double()
    vr.record = None
    zc = vr.sample_coarse(rb).cpu().double()
    zs_helper = torch.sort(torch.cat([zc, zf], -1), -1).values
    print("setup off" if off else "fused setup", ": sorted depths of the differentiable forward vs the helper's: differing",
          int(((zs_fwd - zs_helper).abs() > 1e-7).sum()), "of", zs_fwd.numel(), "max", float((zs_fwd - zs_helper).abs().max()))
    out[off] = dict(zf=zf, rb=rb.cpu(), verts=m.verts.cpu(), o2c=m.ober2cano_transform.cpu(), T=m.verts_transform.cpu(), res={k: v.cpu() for k, v in res.items()})
a, b = out[""], out["1"]
for k in ("rb", "verts", "o2c", "T"):
    print(k, "max abs diff fused vs separate:", float((a[k] - b[k]).abs().max()), "max abs", float(b[k].abs().max()))
print("z_fine differing samples:", int(((a["zf"] - b["zf"]).abs() > 1e-6).sum()), "of", a["zf"].numel(), "max", float((a["zf"] - b["zf"]).abs().max()))
for k in a["res"]:
    print(k, float((a["res"][k] - b["res"][k]).abs().max()))
