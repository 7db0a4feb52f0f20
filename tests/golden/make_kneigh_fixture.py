#!/usr/bin/env python3
"""tests/golden/kneigh.npz: the reference's AnimNeRF with k_neigh = 3 and 6 (every shipped config: 4) — unpose() and
forward() on the seeded synthetic body.  RUNS THE REFERENCE (imported from /root/reference), stores inputs and outputs.

  python tests/golden/make_kneigh_fixture.py
"""
import os, sys, tempfile
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf                       # noqa: E402
from anim_nerf_amd import synthetic as syn       # noqa: E402

SEED = 41
r_anim, _, r_ds = mf.import_reference()
tbl = syn.make_smpl_table(0)
tmp = tempfile.mkdtemp(prefix="anr_smpl_")
os.makedirs(os.path.join(tmp, "smpl"))
tbl.write_pickle(os.path.join(tmp, "smpl", "SMPL_MALE.pkl"))
pose = mf.t(syn.animated_pose_params(seed=1, bs=2))
templ = mf.t(syn.template_pose_params())
c2w_i, foc_i, cen_i = syn.pinhole_camera(4, 4)
rays_w = r_ds.gen_rays(torch.from_numpy(c2w_i), 4, 4, foc_i.tolist(), 0.1, 10.0, cen_i.tolist()).view(1, -1, 8).expand(2, -1, -1).contiguous()
out = {}
for k in (3, 6):
    torch.manual_seed(SEED)
    ref = r_anim.AnimNeRF(model_path=tmp, model_type="smpl", gender="male", freqs_xyz=10, freqs_dir=0, use_view=False,
                          use_unpose=True, k_neigh=k, use_knn=False, use_fine=True, share_fine=False, dis_threshold=0.2).eval()
    mf.sigma_gain_(ref.nerf, 300.0, 2.0)
    with torch.no_grad():
        ref.set_body_model(pose, templ)
        ref.convert_to_body_model_space(rays_w)
        ref.clac_ober2cano_transform()
        if k == 3:
            g = torch.Generator().manual_seed(17)
            n = 1024
            vid = torch.randint(0, syn.NUM_VERTS, (2, n), generator=g)
            scale = torch.tensor([0.01, 0.05, 0.15, 0.4])[torch.randint(0, 4, (2, n, 1), generator=g)]
            xyz = torch.gather(ref.verts, 1, vid[..., None].expand(-1, -1, 3)) + scale * torch.randn(2, n, 3, generator=g)
        xyz_c, _, valid = ref.unpose(xyz)
        rgb, sigma = ref(xyz, None, use_fine=False)
    out.update({f"xyz_c_{k}": xyz_c.numpy(), f"valid_{k}": valid.numpy(), f"rgb_{k}": rgb.numpy(), f"sigma_{k}": sigma.numpy()})
np.savez_compressed(os.path.join(HERE, "kneigh.npz"), seed=SEED, gain=300.0, shift=2.0, xyz=xyz.numpy(), rays_world=rays_w.numpy(), **out)
print("wrote kneigh.npz", {k: v.shape for k, v in out.items()})
