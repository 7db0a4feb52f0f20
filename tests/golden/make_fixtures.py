#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE REFERENCE (imported from /root/reference) on the
seeded synthetic world of anim_nerf_amd.synthetic.  Run in the build container only:

    python tests/golden/make_fixtures.py

The reference itself never enters this repository: the fixtures hold inputs that are cheap to
store, checksums of the inputs that are regenerated from seeds (SMPL-like table, MLP weights),
and the reference's outputs.  tests/test_oracle_golden.py pins the oracle to them;
tests/test_gpu_parity.py compares the HIP path with them on the GPU box.
"""
import hashlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.environ.get("ANIMNERF_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

import anim_nerf_amd as ana                                      # noqa: E402
from anim_nerf_amd import synthetic as syn                        # noqa: E402

torch.set_num_threads(8)


def import_reference():
    cv2 = types.ModuleType("cv2")
    cv2.COLORMAP_JET = 2
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvt.ToTensor = lambda: (lambda x: x)
    tv.transforms = tvt
    sys.modules.setdefault("cv2", cv2)
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    sys.path.insert(0, REF)
    import models.anim_nerf as r_anim                           # noqa: F401
    import models.volume_rendering as r_vr                      # noqa: F401
    import datasets.anim_nerf_dataset as r_ds                   # noqa: F401
    return r_anim, r_vr, r_ds


def sha(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()[:16]


def weights_checksum(net):
    return sha(*[p for _, p in sorted(net.state_dict().items()) if p.dtype == torch.float32])


def t(x):
    return {k: torch.from_numpy(v) for k, v in x.items()}


def sigma_gain_(net, gain, shift):
    """Spread sigma over a useful range (random-init sigma is ~0.017 +- 0.003, SURVEY.md section 7 hard part 4)."""
    with torch.no_grad():
        net.sigma.weight.mul_(gain)
        net.sigma.bias.mul_(gain).add_(shift)


def main():
    r_anim, r_vr, r_ds = import_reference()
    tbl = syn.make_smpl_table(0)
    tmp = tempfile.mkdtemp(prefix="anr_smpl_")
    os.makedirs(os.path.join(tmp, "smpl"))
    tbl.write_pickle(os.path.join(tmp, "smpl", "SMPL_MALE.pkl"))
    meta = dict(table_checksum=syn.table_checksum(tbl))

    def make_ref(seed, use_unpose):
        torch.manual_seed(seed)
        m = r_anim.AnimNeRF(model_path=tmp, model_type="smpl", gender="male", freqs_xyz=10, freqs_dir=0,
                            use_view=False, use_unpose=use_unpose, k_neigh=4, use_knn=False, use_fine=True,
                            share_fine=False, dis_threshold=0.2)
        return m.eval()

    # ---------------------------------------------------------------- a1 rays
    H, W = 9, 13
    c2w = torch.tensor([[0.8, -0.36, 0.48, 0.3], [0.6, 0.48, -0.64, -0.2], [0.0, 0.8, 0.6, 1.5]])
    focal, cen = [17.5, 16.25], [6.25, 4.75]
    rays = r_ds.gen_rays(c2w, H, W, focal, 0.1, 10.0, cen)
    np.savez_compressed(os.path.join(OUT, "rays.npz"), c2w=c2w.numpy(), H=H, W=W, focal=np.float32(focal),
                        center=np.float32(cen), near=0.1, far=10.0, rays=rays.numpy())

    # ---------------------------------------------------------------- a2-a5 per-frame state
    pose = t(syn.animated_pose_params(seed=1, bs=2))
    templ = t(syn.template_pose_params())
    ref = make_ref(7, True)
    with torch.no_grad():
        ref.set_body_model(pose, templ)
        sub = np.arange(0, syn.NUM_VERTS, 53)
        smpl = dict(verts=ref.verts[:, sub], joints=ref.joints, A=ref.joints_transform, T=ref.verts_transform[:, sub],
                    shape_offsets=ref.shape_offsets[:, sub], pose_offsets=ref.pose_offsets[:, sub],
                    verts_template=ref.verts_template[:, sub], T_template=ref.verts_transform_template[:, sub])
        Hc = Wc = 12
        c2w_i, foc_i, cen_i = syn.pinhole_camera(Hc, Wc)
        rays_w = r_ds.gen_rays(torch.from_numpy(c2w_i), Hc, Wc, foc_i.tolist(), 0.1, 10.0, cen_i.tolist())
        rays_w = rays_w.view(1, -1, 8).expand(2, -1, -1).contiguous()
        rays_b = ref.convert_to_body_model_space(rays_w)
        ref.clac_ober2cano_transform()
        frame = dict(rays_world=rays_w, rays_body=rays_b, verts_root=ref.verts[:, sub],
                     T_root=ref.verts_transform[:, sub], global_transform=ref.global_transform,
                     ober2cano=ref.ober2cano_transform[:, sub],
                     ober2cano_sha=np.array(sha(ref.ober2cano_transform)))
    np.savez_compressed(os.path.join(OUT, "frame.npz"), sub=sub,
                        **{f"pose_{k}": v.numpy() for k, v in pose.items()},
                        **{f"smpl_{k}": v.numpy() for k, v in smpl.items()},
                        **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in frame.items()})

    # ---------------------------------------------------------------- a8-a10 warp of explicit points
    with torch.no_grad():
        g = torch.Generator().manual_seed(11)
        n_pts = 3072
        # points around the posed body: vertex + noise of mixed scale (inside / near / far)
        vid = torch.randint(0, syn.NUM_VERTS, (2, n_pts), generator=g)
        scale = torch.tensor([0.01, 0.05, 0.15, 0.4])[torch.randint(0, 4, (2, n_pts, 1), generator=g)]
        xyz = torch.gather(ref.verts, 1, vid[..., None].expand(-1, -1, 3)) + scale * torch.randn(2, n_pts, 3, generator=g)
        dist, Tinv = ref.get_neighbs(xyz, ref.verts, ref.ober2cano_transform.clone())
        xyz_c, _, valid = ref.unpose(xyz)
        d = torch.norm(xyz.unsqueeze(2) - ref.verts.unsqueeze(1), dim=-1)
        nd, ni = d.topk(4, largest=False, dim=-1)
        # K2: unposing the posed vertices returns the template vertices
        xv, _, vv = ref.unpose(ref.verts[:, sub])
    np.savez_compressed(os.path.join(OUT, "warp.npz"), xyz=xyz.numpy(), knn_dist=nd.numpy(), knn_idx=ni.numpy().astype(np.int32),
                        blended_dist=dist.numpy(), xyz_c=xyz_c.numpy(), valid=valid.numpy(),
                        verts_unposed=xv.numpy(), verts_unposed_valid=vv.numpy())

    # ---------------------------------------------------------------- a11-a12 MLP on explicit points
    with torch.no_grad():
        g = torch.Generator().manual_seed(12)
        p = torch.cat([torch.rand(1, 1536, 3, generator=g) * 2 - 1, torch.randn(1, 512, 3, generator=g) * 3], 1)
        rgb_c, sig_c = ref.nerf(p)
        rgb_f, sig_f = ref.nerf_fine(p)
        enc = ref.nerf.encoding_xyz(p[:, :64])
        meta.update(mlp_seed=7, w_coarse=weights_checksum(ref.nerf), w_fine=weights_checksum(ref.nerf_fine))
        # our module, same seed, must initialise to the same weights
        torch.manual_seed(7)
        mine = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_unpose=True, use_fine=True)
        assert weights_checksum(mine.nerf) == meta["w_coarse"] and weights_checksum(mine.nerf_fine) == meta["w_fine"], \
            "seeded init differs from the reference's"
    np.savez_compressed(os.path.join(OUT, "mlp.npz"), xyz=p.numpy(), rgb_coarse=rgb_c.numpy(), sigma_coarse=sig_c.numpy(),
                        rgb_fine=rgb_f.numpy(), sigma_fine=sig_f.numpy(), enc64=enc.numpy())

    # ---------------------------------------------------------------- a6-a15 full render cases
    def render_case(name, use_unpose, n_coarse, n_fine, pose_kind, gain, seed, hw=12, chunk=48):
        ref = make_ref(seed, use_unpose)
        shift = 0.0
        if gain != 1.0:
            with torch.no_grad():
                probe = torch.rand(1, 4096, 3, generator=torch.Generator().manual_seed(5)) * 1.6 - 0.8
                shift_c = -gain * ref.nerf(probe)[1].median().item()
                shift_f = -gain * ref.nerf_fine(probe)[1].median().item()
            sigma_gain_(ref.nerf, gain, shift_c)
            sigma_gain_(ref.nerf_fine, gain, shift_f)
            shift = (shift_c, shift_f)
        pose = t(syn.animated_pose_params(seed=3, bs=1) if pose_kind == "animated" else syn.static_pose_params(bs=1))
        templ = t(syn.template_pose_params())
        c2w_i, foc_i, cen_i = syn.pinhole_camera(hw, hw)
        rays = r_ds.gen_rays(torch.from_numpy(c2w_i), hw, hw, foc_i.tolist(), 0.1, 10.0, cen_i.tolist()).view(1, -1, 8)
        vr = r_vr.VolumeRenderer(n_coarse=n_coarse, n_fine=n_fine, n_fine_depth=0, share_fine=False, white_bkgd=True)
        outs = []
        with torch.no_grad():
            ref.set_body_model(pose, templ)
            rays_b = ref.convert_to_body_model_space(rays)
            ref.clac_ober2cano_transform()
            for i in range(0, rays_b.shape[1], chunk):
                rc = rays_b[:, i:i + chunk]
                o = dict(vr(ref, rc, perturb=0.0))
                # intermediates, by calling the reference's own stage methods
                zc = vr.sample_coarse(rc[..., :8], perturb=0.0)
                w, _, _, _ = vr.composite(ref, rc, zc, coarse=True, far=True, perturb=0.0)
                o.update(z_coarse=zc, weights=w)
                if n_fine > 0:
                    mid = 0.5 * (zc[..., :-1] + zc[..., 1:])
                    zf = vr.sample_fine(mid, w[..., 1:-1].detach(), det=True)
                    zs, _ = torch.sort(torch.cat([zc, zf], -1), -1)
                    w2, _, _, _ = vr.composite(ref, rc, zs, coarse=False, far=True, perturb=0.0)
                    o.update(z_fine=zf, z_sorted=zs, weights_fine=w2)
                outs.append(o)
        res = {k: torch.cat([o[k] for o in outs], 1).numpy() for k in outs[0]}
        np.savez_compressed(
            os.path.join(OUT, f"render_{name}.npz"), rays_world=rays.numpy(), rays_body=rays_b.numpy(),
            n_coarse=n_coarse, n_fine=n_fine, use_unpose=use_unpose, pose_kind=pose_kind, gain=gain,
            shift=np.float64(shift), seed=seed, hw=hw, w_coarse=weights_checksum(ref.nerf),
            w_fine=weights_checksum(ref.nerf_fine), **{f"pose_{k}": v.numpy() for k, v in pose.items()}, **res)
        print(name, {k: (float(v.min()), float(v.max())) for k, v in res.items() if k in ("rgbs_fine", "alphas_fine", "rgbs", "alphas")})

    # BASELINE config 2: 64+64, no warp (literal random init, and a sigma-gain variant that exercises compositing)
    render_case("cfg2_nowarp", False, 64, 64, "static", 1.0, 21)
    render_case("cfg2_nowarp_gain", False, 64, 64, "static", 3000.0, 21)
    # BASELINE config 3: 64+64 with the inverse-LBS / KNN warp, animated pose
    render_case("cfg3_warp_gain", True, 64, 64, "animated", 3000.0, 22)
    # BASELINE config 1 shape: 32 coarse only, warp on
    render_case("cfg1_coarse32_warp", True, 32, 0, "animated", 3000.0, 23)
    # shipped yaml shape: 64 + 32
    render_case("yaml_64_32_warp", True, 64, 32, "animated", 3000.0, 24, hw=8, chunk=32)

    np.savez_compressed(os.path.join(OUT, "meta.npz"), **{k: np.array(v) for k, v in meta.items()})
    print("fixtures written to", OUT, meta)


if __name__ == "__main__":
    main()
