import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from helpers import golden, seeded_model, tdict
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
g = golden("render_cfg3_warp_gain")
m = seeded_model(tbl, g["seed"], True, g["gain"], g["shift"], device=dev, mlp_mode="bf16")
H = W = 1024
c2w, focal, cen = syn.pinhole_camera(H, W)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), H, W, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=100).items()}
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
vr = ana.VolumeRenderer(n_coarse=64, n_fine=64)
idx = (torch.arange(384, 640)[:, None] * W + torch.arange(384, 640)[None]).reshape(-1).to(dev)
crop = rays[:, idx].contiguous()
with torch.no_grad():
    sparse = ana.batched_inference(vr, m, crop, pose, templ, chunk=1 << 16)
    m.skip_far_samples = m.skip_invalid_samples = False
    dense = ana.batched_inference(vr, m, crop, pose, templ, chunk=1 << 14)
    m.skip_far_samples = m.skip_invalid_samples = True
    for k in sparse:
        d = (sparse[k] - dense[k]).abs()
        print(k, "rays differing", int((d.amax(-1) > 0).sum()), "max", d.max().item())
    bad = torch.nonzero((sparse["rgbs_fine"] - dense["rgbs_fine"]).abs().amax(-1)[0] > 0)[:, 0]
    print("bad rays", bad[:10].tolist())
    # stage by stage on the bad rays
    m.set_body_model(pose, templ)
    rb = m.convert_to_body_model_space(crop)
    m.clac_ober2cano_transform()
    sub = rb[:, bad[:64]].contiguous()
    zc = vr.sample_coarse(sub)
    lean = m.warped_points(rays=sub, z=zc, skip_far=True, lean=True)
    m.skip_far_samples = False
    exact = m.warped_points(rays=sub, z=zc, skip_far=False)
    m.skip_far_samples = True
    vl, ve = lean[1].view(-1), exact[:, 3]
    print("coarse validity mismatches", int((vl.float() != ve).sum()), "of", vl.numel())
    both = (vl > 0) & (ve > 0)
    print("coarse pts max diff on valid", (lean[0][both][:, :3] - exact[both][:, :3]).abs().max().item())
