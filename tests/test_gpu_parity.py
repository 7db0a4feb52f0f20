"""GPU parity tests: every HIP entry point (through the C ABI, via anim_nerf_amd.ops / the module
classes) against the CPU oracle on the same seeded inputs and against the reference's golden
vectors.  Tolerance on rendered RGB / sigma: 1e-4 relative (BASELINE.json north_star), fp32 mode.
Run with:  python -m pytest tests -m gpu
"""
import os

import numpy as np
import pytest
import torch

from helpers import weights_checksum, golden, net_params, oracle_table, rel_err, seeded_model, tdict
from oracle import animnerf_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    import anim_nerf_amd as ana
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ana._lib.load()
    return torch.device("cuda:0")


def _templ(device=None):
    from anim_nerf_amd import synthetic as syn
    return {k: torch.from_numpy(v).to(device) if device else torch.from_numpy(v) for k, v in syn.template_pose_params().items()}


def _to(d, device):
    return {k: v.to(device) for k, v in d.items()}


# ----------------------------------------------------------------------------- a1
def test_ray_gen(dev):
    import anim_nerf_amd as ana
    g = golden("rays")
    rays = ana.gen_rays(torch.from_numpy(g["c2w"]).to(dev), int(g["H"]), int(g["W"]), g["focal"].tolist(),
                        float(g["near"]), float(g["far"]), g["center"].tolist())
    torch.testing.assert_close(rays.cpu(), torch.from_numpy(g["rays"]), rtol=1e-6, atol=1e-6)
    # full BASELINE size: unit directions, constant origin
    c2w, focal, cen = ana.synthetic.pinhole_camera(1024, 1024)
    big = ana.gen_rays(torch.from_numpy(c2w).to(dev), 1024, 1024, focal.tolist(), 0.1, 10.0, cen.tolist())
    assert big.shape == (1024, 1024, 8)
    assert (big[..., 3:6].norm(dim=-1) - 1).abs().max() < 1e-6 and big[..., :3].abs().max() == 0
    ref = orc.make_rays(torch.from_numpy(c2w), 1024, 1024, focal.tolist(), 0.1, 10.0, cen.tolist())
    torch.testing.assert_close(big.cpu(), ref, rtol=1e-6, atol=1e-6)


# ----------------------------------------------------------------------------- a2-a5
def test_frame_state(dev, smpl_table):
    """set_body_model / convert_to_body_model_space / clac_ober2cano_transform vs the reference's outputs."""
    g = golden("frame")
    m = seeded_model(smpl_table, 7, True, device=dev)
    m.set_body_model(_to(tdict(g), dev), _templ(dev))
    sub = torch.from_numpy(g["sub"]).to(dev)
    torch.testing.assert_close(m.verts[:, sub].cpu(), torch.from_numpy(g["smpl_verts"]), rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(m.verts_transform[:, sub].cpu(), torch.from_numpy(g["smpl_T"]), rtol=1e-5, atol=5e-6)
    rays_b = m.convert_to_body_model_space(torch.from_numpy(g["rays_world"]).to(dev))
    torch.testing.assert_close(rays_b.cpu(), torch.from_numpy(g["rays_body"]), rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(m.verts[:, sub].cpu(), torch.from_numpy(g["verts_root"]), rtol=1e-5, atol=5e-6)
    m.clac_ober2cano_transform()
    torch.testing.assert_close(m.ober2cano_transform[:, sub].cpu(), torch.from_numpy(g["ober2cano"]), rtol=1e-4, atol=1e-5)
    # K1: ober2cano[v] . verts_posed[v] == verts_template[v]; K8: affine last row
    import anim_nerf_amd as ana
    back = ana.batch_transform(m.ober2cano_transform, m.verts)
    assert (back - m.verts_template).abs().max() < 1e-5
    last = m.ober2cano_transform[..., 3, :]
    assert torch.equal(last, torch.tensor([0., 0, 0, 1], device=dev).expand_as(last))


def test_fused_frame_setup_matches_reference_and_the_separate_kernels(dev, smpl_table):
    """AnimNeRF.frame_setup / ops.frame_setup (csrc/frame_setup.hip: the per-frame set-up in two launches, rest joints as
    J0 + JS . betas) against the REFERENCE's outputs (tests/golden/frame.npz, the gates of test_frame_state), against the
    eight separate kernels on a batch of frames (fp32 rounding: a different summation order, nothing else), from per-frame
    arrays and from the BodyModelParams tables with a frame index (a row used twice, the shared betas row)."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import ops, synthetic as syn
    g = golden("frame")
    m = seeded_model(smpl_table, 7, True, device=dev)
    sub = torch.from_numpy(g["sub"]).to(dev)
    with torch.no_grad():
        rays_b = m.frame_setup(_to(tdict(g), dev), _templ(dev), torch.from_numpy(g["rays_world"]).to(dev))
    torch.testing.assert_close(rays_b.cpu(), torch.from_numpy(g["rays_body"]), rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(m.verts[:, sub].cpu(), torch.from_numpy(g["verts_root"]), rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(m.ober2cano_transform[:, sub].cpu(), torch.from_numpy(g["ober2cano"]), rtol=1e-4, atol=1e-5)
    back = ana.batch_transform(m.ober2cano_transform, m.verts)
    assert (back - m.verts_template).abs().max() < 1e-5
    # a batch of frames: the fused launches against the separate kernels
    F = 5
    seeded = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=31, bs=F).items()}
    seeded["betas"] = seeded["betas"] + 0.3 * torch.randn(seeded["betas"].shape, generator=torch.Generator().manual_seed(2)).to(dev)
    c2w, focal, cen = syn.pinhole_camera(12, 12)
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 12, 12, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8).repeat(F, 1, 1).contiguous()
    a = seeded_model(smpl_table, 7, True, device=dev)
    with torch.no_grad():
        a.set_body_model(seeded, _templ(dev))
        A_sep, so_sep, po_sep = a.joints_transform, a.shape_offsets, a.pose_offsets
        rays_sep = a.convert_to_body_model_space(rays)
        a.clac_ober2cano_transform()
        b = seeded_model(smpl_table, 7, True, device=dev)
        rays_fused = b.frame_setup(seeded, _templ(dev), rays)
    pairs = [("rays", rays_fused, rays_sep), ("verts", b.verts, a.verts), ("joints", b.joints, a.joints), ("A", b.joints_transform, A_sep),
             ("T", b.verts_transform, a.verts_transform), ("global", b.global_transform, a.global_transform),
             ("shape offsets", b.shape_offsets, so_sep), ("pose offsets", b.pose_offsets, po_sep), ("ober2cano", b.ober2cano_transform, a.ober2cano_transform)]
    for name, x, y in pairs:
        assert x.shape == y.shape, name
        assert (x - y).abs().max() <= 2e-6 * max(1.0, float(y.abs().max())), (name, float((x - y).abs().max()))
    # tables + a frame index: rows gathered by the kernel (a row used twice; ONE betas row for all frames)
    table = ana.BodyModelParams(9).to(dev)
    rows = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=40, bs=9).items()}
    for name in table.param_names:
        table.init_parameters(name, rows[name])
    fidx = torch.tensor([3, 8, 3, 0], device=dev)
    w = {n: getattr(table, n).weight for n in table.param_names}
    T = (b.verts_transform_template, b.shape_offsets_template, b.pose_offsets_template)
    o_tab = ops.frame_setup((w["betas"], w["global_orient"], w["body_pose"], w["transl"]), fidx, b._chain_consts(), b.body_model, T, rays[:4])
    per = {n: (w[n][torch.zeros_like(fidx)] if n == "betas" else w[n][fidx]).contiguous() for n in table.param_names}
    o_arr = ops.frame_setup((per["betas"], per["global_orient"], per["body_pose"], per["transl"]), None, b._chain_consts(), b.body_model, T, rays[:4])
    for k in o_tab:
        assert torch.equal(o_tab[k], o_arr[k]), k
    assert torch.equal(o_tab["pose"], torch.cat([per["global_orient"], per["body_pose"]], 1)) and torch.equal(o_tab["verts"][0], o_tab["verts"][2])
    # a frame index outside the tables (nn.Embedding raises there): nothing is read out of bounds, that frame — and only that
    # frame — comes out NaN, so the step's loss says so at the next host read
    bad = torch.tensor([3, 9, -1, 0], device=dev)
    o_bad = ops.frame_setup((w["betas"], w["global_orient"], w["body_pose"], w["transl"]), bad, b._chain_consts(), b.body_model, T, rays[:4])
    for k in ("pose", "A", "verts", "ober2cano", "rays_body"):
        assert torch.equal(o_bad[k][0], o_tab[k][0]) and torch.equal(o_bad[k][3], o_tab[k][3]), k
        assert torch.isnan(o_bad[k][1]).any() and torch.isnan(o_bad[k][2]).any(), k
    assert torch.isnan(o_bad["pose"][1:3]).all() and torch.isnan(o_bad["verts"][1:3]).all()


class _one_body:
    """View of a posed model restricted to body b (what a bs = 1 checker needs)."""

    def __init__(self, m, b):
        self._m, self.verts, self.ober2cano_transform = m, m.verts[b:b + 1].contiguous(), m.ober2cano_transform[b:b + 1].contiguous()
        self._index = m.knn_index()[b:b + 1].contiguous()

    def knn_index(self):
        return self._index

    def __getattr__(self, k):
        return getattr(self._m, k)


def _warp_frame(dev, smpl_table):
    g = golden("frame")
    m = seeded_model(smpl_table, 7, True, device=dev)
    m.set_body_model(_to(tdict(g), dev), _templ(dev))
    m.convert_to_body_model_space(torch.from_numpy(g["rays_world"]).to(dev))
    m.clac_ober2cano_transform()
    return m


# ----------------------------------------------------------------------------- a8-a10
def test_knn_matches_reference(dev, smpl_table):
    import anim_nerf_amd as ana
    g = golden("warp")
    m = _warp_frame(dev, smpl_table)
    xyz = torch.from_numpy(g["xyz"]).to(dev)
    dist, idx = ana.ops.knn(m.verts, xyz)
    assert idx.dtype == torch.int64 and dist.shape == (2, xyz.shape[1], 4)
    d_ref, i_ref = torch.from_numpy(g["knn_dist"]), torch.from_numpy(g["knn_idx"]).long()
    torch.testing.assert_close(dist.cpu(), d_ref, rtol=1e-5, atol=1e-6)
    same = (idx.cpu() == i_ref)
    # index differences are allowed only between (near-)tied distances
    tied = (dist.cpu() - d_ref).abs() <= 1e-6
    assert (same | tied).all()
    assert (dist[..., 1:] >= dist[..., :-1]).all()          # ascending


def test_warp_matches_reference(dev, smpl_table):
    import anim_nerf_amd as ana
    g = golden("warp")
    m = _warp_frame(dev, smpl_table)
    xyz = torch.from_numpy(g["xyz"]).to(dev)
    pts, dist, idx, blended = ana.ops.warp_points(m.knn_index(), m.ober2cano_transform, m.body_model.lbs_weights, 0.2,
                                                  xyz=xyz, debug=True)
    b_ref = torch.from_numpy(g["blended_dist"])[..., 0]
    ok = (blended.cpu() - b_ref).abs() <= 1e-5 + 1e-5 * b_ref.abs()
    from accounting import neighbour_discontinuity
    lbs = oracle_table(smpl_table)["lbs_weights"]
    excuse = neighbour_discontinuity(lbs, torch.from_numpy(g["knn_dist"]), torch.from_numpy(g["knn_idx"]).long())
    assert (ok | excuse).all(), "blended distance differs away from a neighbour tie / confidence threshold"
    assert (~ok).sum() <= 16, "sanity: those are a handful of points"
    v_ref = torch.from_numpy(g["valid"])[..., 0]
    v_ok = (pts[..., 3].cpu() == v_ref) | ((b_ref - 0.2).abs() < 1e-5)
    assert v_ok[ok].all()
    x_ref = torch.from_numpy(g["xyz_c"])
    err = (pts[..., :3].cpu() - x_ref).abs().max(-1).values
    assert (err[ok] <= 1e-5 + RTOL * x_ref.abs().max(-1).values[ok]).all()
    # module API: unpose() and forward()
    xc, _, valid = m.unpose(xyz)
    assert torch.equal(xc, pts[..., :3]) and torch.equal(valid[..., 0], pts[..., 3])
    # K2 on the reference's output for posed vertices
    sub = torch.from_numpy(golden("frame")["sub"]).to(dev)
    xv, _, vv = m.unpose(m.verts[:, sub].contiguous())
    torch.testing.assert_close(xv.cpu(), torch.from_numpy(g["verts_unposed"]), rtol=1e-4, atol=2e-5)
    assert vv.min() == 1


def test_warp_from_rays_equals_explicit_points(dev, smpl_table):
    import anim_nerf_amd as ana
    m = _warp_frame(dev, smpl_table)
    g = golden("frame")
    rays = torch.from_numpy(g["rays_body"]).to(dev)
    vr = ana.VolumeRenderer(n_coarse=16)
    z = vr.sample_coarse(rays)
    a = ana.ops.warp_points(m.knn_index(), m.ober2cano_transform, m.body_model.lbs_weights, 0.2, rays=rays, z=z)
    xyz = (rays[..., None, :3] + z[..., None] * rays[..., None, 3:6]).reshape(2, -1, 3)
    b = ana.ops.warp_points(m.knn_index(), m.ober2cano_transform, m.body_model.lbs_weights, 0.2, xyz=xyz)
    # torch may contract o + z*d into an fma (1 ulp); the kernel rounds product and sum separately
    # against the oracle; a point may differ only at one of the reference's discontinuities: the validity threshold, a
    # neighbour tie, a confidence threshold
    from accounting import neighbour_discontinuity
    xc, valid, dbg = orc.warp_to_canonical(xyz.cpu(), m.verts.cpu(), m.body_model.lbs_weights.cpu(),
                                           m.ober2cano_transform.cpu(), 0.2, chunk=1024)
    tie = neighbour_discontinuity(m.body_model.lbs_weights.cpu(), dbg["dist"], dbg["idx"])
    edge = (dbg["blended"][..., 0] - 0.2).abs() <= 1e-5
    a_, b_ = a.cpu().view(2, -1, 4), b.cpu().view(2, -1, 4)
    assert ((a_[..., 3] == b_[..., 3]) | edge | tie).all()
    assert (((a_[..., :3] - b_[..., :3]).abs().max(-1).values < 2e-5) | tie).all()
    assert ((a_[..., 3] == valid[..., 0]) | edge | tie).all()
    both = (a_[..., 3] >= 1) & (valid[..., 0] >= 1)
    assert (((a_[..., :3] - xc).abs().max(-1).values < 1e-5 + RTOL * xc.abs().max(-1).values) | tie | ~both).all()


def test_warp_two_pass_equals_one_pass(dev, smpl_table):
    """Renderer mode (skip_far): classify + bin + search-the-list gives, on every valid sample, the bits of the one-pass
    kernel and of the exact search everywhere, and the same valid flags; bs = 2, rays and explicit points, neighbour
    outputs.  (What is stored in xyz of an INVALID sample — the raw position or its warp — is not part of the contract:
    nothing consumes it.)"""
    import anim_nerf_amd as ana
    m = _warp_frame(dev, smpl_table)
    g = golden("frame")
    rays = torch.from_numpy(g["rays_body"]).to(dev)
    args = (m.knn_index(), m.ober2cano_transform, m.body_model.lbs_weights, 0.2)
    for K in (16, 33):
        z = ana.VolumeRenderer(n_coarse=K).sample_coarse(rays)
        one = ana.ops.warp_points(*args, rays=rays, z=z, skip_far=True, neighbours=True, two_pass=False)
        two = ana.ops.warp_points(*args, rays=rays, z=z, skip_far=True, neighbours=True, two_pass=True)
        exact = ana.ops.warp_points(*args, rays=rays, z=z, skip_far=False)
        v = exact[..., 3] == 1
        assert 0.01 < v.float().mean() < 0.9
        for got in (one, two):
            assert torch.equal(got[0][..., 3], exact[..., 3]) and torch.equal(got[0][v], exact[v])
        for a, b in zip(one, two):
            assert torch.equal(a[v], b[v])
    xyz = (rays[..., None, :3] + z[..., None] * rays[..., None, 3:6]).reshape(2, -1, 3).contiguous()
    one = ana.ops.warp_points(*args, xyz=xyz, skip_far=True, two_pass=False)
    two = ana.ops.warp_points(*args, xyz=xyz, skip_far=True, two_pass=True)
    assert torch.equal(one[..., 3], two[..., 3]) and torch.equal(one[one[..., 3] == 1], two[two[..., 3] == 1])


def test_reach_mask_drops_only_samples_no_vertex_can_reach(dev, smpl_table):
    """The index built with the reach mask (anr_knn_index_build_reach: a 32^3 grid of "some vertex within dis_threshold of this
    cell") against the plain index: the classify pass of anr_warp_points lists FEWER samples for the search, and every output a
    consumer reads — validity, canonical points, neighbour ids and blend weights of the valid samples, the lean list — carries
    the same bits; training-shaped batch and a dense one (the cell-sorted path); a radius above the mask's falls back to the
    box test."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import ops, synthetic as syn
    m = seeded_model(smpl_table, 3, True, device=dev)
    for bs, hw, n_rays, K in ((3, 128, 1024, 64), (1, 256, 256 * 256, 16)):
        pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=40 + bs, bs=bs, pose_std=0.4).items()}
        templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
        c2w, focal, cen = syn.pinhole_camera(hw, hw)
        full = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(-1, 8)
        gen = torch.Generator().manual_seed(5)
        pick = torch.stack([torch.randperm(hw * hw, generator=gen)[:n_rays] for _ in range(bs)]).to(dev)
        with torch.no_grad():
            m.set_body_model(pose, templ)
            rays = m.convert_to_body_model_space(full[pick].contiguous())
            m.clac_ober2cano_transform()
            z = ana.VolumeRenderer(n_coarse=K).sample_coarse(rays)
            plain = ops.knn_index_build(m.verts, m.knn_order)
            masked = ops.knn_index_build(m.verts, m.knn_order, reach=0.2)
            nfl = m.knn_index().shape[1]
            assert plain.shape == masked.shape and masked.shape[1] == nfl
            bits = masked[:, -4096:].contiguous().view(torch.int32)
            set_frac = sum(bin(int(w) & 0xffffffff).count("1") for w in bits.flatten().tolist()) / (bs * 32768)
            assert 0.02 < set_frac < 0.6, set_frac                 # a body fills a small part of its padded box
            # the tree itself is the same: everything but the mask (last 4 KB) and the radius in the body box's 4th float
            Vp = -(-m.verts.shape[1] // 8) * 8
            box3 = nfl - 4096 - 4 * Vp - 32 + 12
            assert torch.equal(plain[:, :box3], masked[:, :box3]) and torch.equal(plain[:, box3 + 4:-4096], masked[:, box3 + 4:-4096])
            assert masked[:, box3:box3 + 4].contiguous().view(torch.float32).eq(0.2).all() and not plain[:, box3:box3 + 4].any()
            rest = (m.ober2cano_transform, m.body_model.lbs_weights)
            for thr in (0.2, 0.1, 0.25):                            # 0.25 > the mask's radius: box test only
                a = ops.warp_points(plain, *rest, thr, rays=rays, z=z, skip_far=True, neighbours=True)
                b = ops.warp_points(masked, *rest, thr, rays=rays, z=z, skip_far=True, neighbours=True)
                v = a[0][..., 3] == 1
                assert torch.equal(a[0][..., 3], b[0][..., 3]) and 0.005 < v.float().mean() < 0.9
                for x, y in zip(a, b):
                    assert torch.equal(x[v], y[v])
                la = ops.warp_points(plain, *rest, thr, rays=rays, z=z, skip_far=True, lean=True)
                lb = ops.warp_points(masked, *rest, thr, rays=rays, z=z, skip_far=True, lean=True)
                assert torch.equal(la[1], lb[1]) and torch.equal(la[3], lb[3])
                n = int(la[3].item())
                # (the list's order is the order its workgroups reached the counter in: compared as a set)
                assert torch.equal(torch.sort(la[2][:n])[0], torch.sort(lb[2][:n])[0]) and torch.equal(la[0][la[1] == 1], lb[0][lb[1] == 1])


@pytest.mark.parametrize("bs,n_rays,K", [(4, 1024, 64), (1, 77, 20), (3, 300, 33)])
def test_warp_small_batch_group_search_is_bit_identical(dev, smpl_table, bs, n_rays, K, monkeypatch):
    """A training-shaped batch (random pixels, a few bodies in different poses) goes through warp_search_groups_kernel
    (eight lanes per sample): on every valid sample the canonical point, the neighbour ids and the blend weights carry the
    bits of the lane-per-sample search of the same list (ANR_WARP_LANE_PER_SAMPLE) and of the exact search of every
    sample; valid flags identical everywhere; the lean outputs (validity bytes, list of valid samples) agree too."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    m = seeded_model(smpl_table, 3, True, device=dev)
    pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=90 + bs, bs=bs, pose_std=0.4).items()}
    hw = 128
    c2w, focal, cen = syn.pinhole_camera(hw, hw)
    full = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(-1, 8)
    gen = torch.Generator().manual_seed(bs)
    pick = torch.stack([torch.randperm(hw * hw, generator=gen)[:n_rays] for _ in range(bs)]).to(dev)
    with torch.no_grad():
        m.set_body_model(pose, _templ(dev))
        rays = m.convert_to_body_model_space(full[pick].contiguous())
        m.clac_ober2cano_transform()
        z = ana.VolumeRenderer(n_coarse=K).sample_coarse(rays)
        args = (m.knn_index(), m.ober2cano_transform, m.body_model.lbs_weights, 0.2)
        groups = ana.ops.warp_points(*args, rays=rays, z=z, skip_far=True, neighbours=True, two_pass=True)
        lean_g = ana.ops.warp_points(*args, rays=rays, z=z, skip_far=True, lean=True)
        monkeypatch.setenv("ANR_WARP_LANE_PER_SAMPLE", "1")
        lanes = ana.ops.warp_points(*args, rays=rays, z=z, skip_far=True, neighbours=True, two_pass=True)
        lean_l = ana.ops.warp_points(*args, rays=rays, z=z, skip_far=True, lean=True)
        monkeypatch.delenv("ANR_WARP_LANE_PER_SAMPLE")
        exact = ana.ops.warp_points(*args, rays=rays, z=z, skip_far=False, neighbours=True)
    v = exact[0][..., 3] == 1
    assert 0.01 < v.float().mean() < 0.9
    for got in (groups, lanes):
        assert torch.equal(got[0][..., 3], exact[0][..., 3])
        for a, b in zip(got, exact):
            assert torch.equal(a[v], b[v])
    for lean in (lean_g, lean_l):                       # (pts, validity bytes, list of valid positions, their count)
        assert torch.equal(lean[1].bool(), v) and torch.equal(lean[0][v], exact[0][v])
        n_valid = int(lean[3].item())
        assert n_valid == int(v.sum()) and torch.equal(lean[2][:n_valid].long().sort().values, v.view(-1).nonzero()[:, 0])


def test_fine_call_copies_the_coarse_calls_cell_results(dev, smpl_table, monkeypatch):
    """anr_warp_points_cells (round 6): the lean fine call of a frame takes the per-cell radii / seeds / dead flags out of the
    coarse call's workspace (it rides on the coarse validity bytes, ops.warp_points) instead of searching those cells again.
    Same validity bytes, same canonical points on the valid samples, same valid list as the call that searches every cell
    itself (ANR_WARP_NO_PREV_CELLS) and as the exact search of every sample — with fine depths that reach cells the coarse
    samples did not touch, and with no sample copied (perm = 255 everywhere: every fine sample is searched)."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    m = seeded_model(smpl_table, 3, True, device=dev)
    pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=17, bs=1, pose_std=0.3).items()}
    hw = 128
    c2w, focal, cen = syn.pinhole_camera(hw, hw)
    full = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
    with torch.no_grad():
        m.set_body_model(pose, _templ(dev))
        rays = m.convert_to_body_model_space(full.contiguous())
        m.clac_ober2cano_transform()
        zc = ana.VolumeRenderer(n_coarse=64).sample_coarse(rays)                 # 16,384 x 64 = 2^20 samples: the cell pass
        args = (m.knn_index(), m.ober2cano_transform, m.body_model.lbs_weights, 0.2)
        coarse = ana.ops.warp_points(*args, rays=rays, z=zc, skip_far=True, lean=True)
        assert coarse[1].warp_cells[1] == zc.shape[1] * 64
        # "fine" depths: 128 per ray, between and beyond the coarse ones
        g = torch.Generator(device="cpu").manual_seed(3)
        zf = (zc[..., :1] + (zc[..., -1:] - zc[..., :1]) * torch.rand(1, zc.shape[1], 128, generator=g).to(dev) * 1.02).sort(-1).values
        perm = torch.full((1, zc.shape[1] * 128), 255, dtype=torch.uint8, device=dev)
        with_prev = ana.ops.warp_points(*args, rays=rays, z=zf, skip_far=True, lean=True, reuse=(coarse[0], coarse[1], perm))
        monkeypatch.setenv("ANR_WARP_NO_PREV_CELLS", "1")
        without = ana.ops.warp_points(*args, rays=rays, z=zf, skip_far=True, lean=True, reuse=(coarse[0], coarse[1], perm))
        monkeypatch.delenv("ANR_WARP_NO_PREV_CELLS")
        exact = ana.ops.warp_points(*args, rays=rays, z=zf, skip_far=False)
    v = exact[..., 3] == 1
    assert 0.01 < v.float().mean() < 0.9
    for got in (with_prev, without):
        assert torch.equal(got[1].bool(), v) and torch.equal(got[0][v], exact[v])
        n_valid = int(got[3].item())
        assert n_valid == int(v.sum()) and torch.equal(got[2][:n_valid].long().sort().values, v.view(-1).nonzero()[:, 0])


# ----------------------------------------------------------------------------- a11-a12
@pytest.mark.parametrize("flag", [0, 0x100], ids=["lds_dma", "reg_staged"])
def test_mlp_fp32_matches_reference(dev, smpl_table, flag):
    import anim_nerf_amd as ana
    g, meta = golden("mlp"), golden("meta")
    m = seeded_model(smpl_table, int(meta["mlp_seed"]), True, device=dev)
    xyz = torch.from_numpy(g["xyz"]).to(dev)[0]
    pts = torch.cat([xyz, torch.ones_like(xyz[:, :1])], -1)
    pts[5, 3] = 0.0                                            # one invalid point -> sigma = -1e5
    for net, tag in ((m.nerf, "coarse"), (m.nerf_fine, "fine")):
        pack, mode = net.weight_pack("f32")
        out = ana.ops.mlp_forward(pack, mode | flag, pts).cpu()
        rgb_ref, sig_ref = torch.from_numpy(g["rgb_" + tag])[0], torch.from_numpy(g["sigma_" + tag])[0, :, 0]
        assert out[5, 3] == -1e5
        keep = torch.arange(len(xyz)) != 5
        assert rel_err(out[keep, :3], rgb_ref[keep]) < RTOL
        assert ((out[keep, 3] - sig_ref[keep]).abs() <= RTOL * sig_ref[keep].abs() + 1e-4 * sig_ref.abs().median()).all()


def test_mlp_bf16_close_to_fp32_and_tail_sizes(dev, smpl_table):
    import anim_nerf_amd as ana
    m = seeded_model(smpl_table, 7, True, device=dev)
    gen = torch.Generator().manual_seed(9)
    for n in (1, 31, 129, 1000):                                # ragged sizes: partial waves / partial workgroups
        xyz = (torch.rand(n, 3, generator=gen) * 2 - 1).to(dev)
        pts = torch.cat([xyz, torch.ones_like(xyz[:, :1])], -1)
        f32 = m.nerf.eval_points(pts, "f32")
        b16 = m.nerf.eval_points(pts, "bf16")
        rgb_o, sig_o = orc.mlp_forward(net_params(m.nerf), xyz.cpu())
        assert rel_err(f32[:, :3].cpu(), rgb_o) < RTOL and (f32[:, 3].cpu() - sig_o[:, 0]).abs().max() < 2e-6
        assert (b16[:, :3] - f32[:, :3]).abs().max() < 2e-2 and (b16[:, 3] - f32[:, 3]).abs().max() < 5e-3


def test_animnerf_forward_api(dev, smpl_table):
    """AnimNeRF.forward(xyz) -> (rgb, sigma) as the mesh-extraction loop calls it (extract_mesh.py:49-61)."""
    import anim_nerf_amd as ana
    m = _warp_frame(dev, smpl_table)
    g = golden("warp")
    xyz = torch.from_numpy(g["xyz"]).to(dev)[:, :512].contiguous()
    rgb, sigma = m(xyz, None, use_fine=True)
    assert rgb.shape == (2, 512, 3) and sigma.shape == (2, 512, 1)
    from accounting import account_for_points
    for b in range(2):                                          # point by point: within 1e-4 or a named discontinuity
        mb = _one_body(m, b)
        st = account_for_points(mb, oracle_table(smpl_table), xyz[b:b + 1], sigma[b:b + 1], rgb[b:b + 1], use_fine=True, label=f"forward api, body {b}")
        assert st["valid"] > 50
    sg = ana.sigma_grid_inference(m, xyz, chunk=200)
    assert torch.equal(sg, torch.relu(sigma))


# ----------------------------------------------------------------------------- a6, a13, a14
def test_sampling_and_compositing_kernels(dev):
    import anim_nerf_amd as ana
    gen = torch.Generator().manual_seed(4)
    R = 777                                                     # not a multiple of the 4 rays per workgroup
    rays = torch.zeros(1, R, 8)
    rays[..., 3:6] = torch.nn.functional.normalize(torch.randn(1, R, 3, generator=gen), dim=-1)
    rays[..., :3] = torch.randn(1, R, 3, generator=gen)
    rays[..., 6] = 1.5 + torch.rand(1, R, generator=gen)
    rays[..., 7] = 3.5 + torch.rand(1, R, generator=gen)
    for Kc, Kf in ((64, 64), (64, 32), (32, 16), (8, 5), (128, 128)):
        vr = ana.VolumeRenderer(n_coarse=Kc, n_fine=Kf)
        z = vr.sample_coarse(rays.to(dev))
        z_o = orc.coarse_depths(rays, Kc)
        assert torch.equal(z.cpu(), z_o), "coarse depths must be bit-identical"
        rgbs = torch.rand(R, Kc, 3, generator=gen)
        sig = torch.randn(R, Kc, generator=gen) * 20
        sig[::7] = -1e5                                         # K5: empty rays
        sig[3::7, -1] = 5.0                                     # K6: opaque tail
        w_o, c_o, d_o, a_o = orc.composite(rgbs, sig, z_o[0], rays[0, :, 7:8])
        packed = torch.cat([rgbs, sig[..., None]], -1).to(dev)
        w, c, d, a = ana.ops.composite(packed, z[0], rays[0].to(dev), True)
        torch.testing.assert_close(w.cpu(), w_o, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(c.cpu(), c_o, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(d.cpu(), d_o, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(a.cpu(), a_o, rtol=1e-5, atol=1e-6)
        assert torch.equal(c[::7].cpu(), torch.ones_like(c[::7].cpu())) and a[::7].abs().max() == 0      # K5
        assert (d[::7, 0].cpu() == rays[0, ::7, 7]).all()
        # importance sampling + merge
        zs, zf = ana.ops.sample_fine_merge(z[0], w, vr._table(dev, "u", Kf), want_fine=True)
        zf_o = orc.fine_depths(z_o[0], w_o, Kf)
        # The reference replaces cdf differences below eps = 1e-5 by 1 (volume_rendering.py:92-93): in bins whose
        # pdf is ~1e-5 (empty space once the ray is opaque) that branch flips with the last ulp of the cdf and
        # moves the sample by up to one bin.  Compare only samples that fall in bins with pdf clear of eps;
        # the others carry no weight and are bounded by one bin width.
        wq = w_o[:, 1:-1] + 1e-5
        pdf = wq / wq.sum(-1, keepdim=True)
        cdf = torch.cat([torch.zeros(R, 1), torch.cumsum(pdf, -1)], -1)
        u = torch.linspace(0., 1., Kf).expand(R, Kf).contiguous()
        hi = torch.searchsorted(cdf, u, right=True).clamp(1, Kc - 2)
        den = torch.gather(cdf, -1, hi) - torch.gather(cdf, -1, hi - 1)
        solid = (den > 3e-5) & (u < 1.0)
        assert solid.float().mean() > 0.5
        assert ((zf.cpu() - zf_o).abs()[solid] < 2e-5).all()
        width = (rays[0, :, 7] - rays[0, :, 6])[:, None] / Kc
        assert ((zf.cpu() - zf_o).abs() <= 1.01 * width + 2e-5).all()
        assert (zs[:, 1:] >= zs[:, :-1]).all()                  # sortedness
        # merge is an exact permutation of its inputs
        assert torch.equal(torch.sort(torch.cat([z[0], zf], -1), -1).values, zs)
        # random u (training path): still an exact sort
        u = torch.rand(R, Kf, generator=gen).to(dev)
        zs2, zf2 = ana.ops.sample_fine_merge(z[0], w, u, want_fine=True)
        assert torch.equal(torch.sort(torch.cat([z[0], zf2], -1), -1).values, zs2)
        from accounting import importance_sample_excuse
        zf2_o, det2 = orc.fine_depths(z_o[0], w_o, Kf, u=u.cpu(), details=True)
        off, excuse = importance_sample_excuse(zf2.cpu(), zf2_o, det2)    # off only at the reference's own discontinuities
        assert (off <= excuse).all(), int((off & ~excuse).sum())


def test_fused_coarse_pass_equals_composite_then_merge(dev, smpl_table):
    """anr_composite_sample (coarse compositing + importance sampling + merge in one launch, depths from the step table or
    from an array) returns the bits of anr_composite followed by anr_sample_fine_merge, for every shape class of the
    dispatch, ragged ray counts, validity bytes and per-ray uniforms; anr_mlp_forward_rays_steps (depths computed in the
    MLP kernel) returns the bits of anr_sample_coarse + anr_mlp_forward_rays."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import ops
    gen = torch.Generator().manual_seed(8)
    for R, Kc, Kf in ((777, 64, 64), (5, 64, 32), (1030, 32, 16), (333, 8, 5), (257, 100, 28), (130, 64, 128), (99, 128, 128),
                      (61, 160, 64), (40, 200, 56)):
        rays = torch.zeros(R, 8)
        rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
        rays[:, 6] = 1.5 + torch.rand(R, generator=gen)
        rays[:, 7] = 3.5 + torch.rand(R, generator=gen)
        rays = rays.to(dev)
        vr = ana.VolumeRenderer(n_coarse=Kc, n_fine=Kf)
        steps, u = vr._table(dev, "steps", Kc), vr._table(dev, "u", Kf)
        z = ops.sample_coarse(rays, steps)
        rgbs = torch.cat([torch.rand(R, Kc, 3, generator=gen), torch.randn(R, Kc, 1, generator=gen) * 20], -1)
        rgbs[::7, :, 3] = -1e5
        rgbs = rgbs.to(dev)
        valid = (torch.rand(R, Kc, generator=gen) < 0.6).to(torch.uint8).to(dev)
        for vmask in (None, valid):
            for uu in (u, torch.rand(R, Kf, generator=gen).to(dev)):
                w, c, d, a = ops.composite(rgbs, z, rays, True, valid=vmask)
                zs, zf, perm = ops.sample_fine_merge(z, w, uu, want_fine=True, want_perm=True, perm_u8=True)
                for kw in (dict(z=z), dict(steps=steps)):
                    f = ops.composite_sample(rgbs, rays, uu, True, valid=vmask, want_weights=True, want_fine=True,
                                             want_perm=True, **kw)
                    tag = (R, Kc, Kf, vmask is not None, uu.dim(), list(kw))
                    for name, want in (("weights", w), ("rgb", c), ("depth", d), ("acc", a), ("z_sorted", zs), ("z_fine", zf),
                                       ("perm", perm)):
                        assert torch.equal(f[name], want), (name, tag)
                lean = ops.composite_sample(rgbs, rays, uu, True, valid=vmask, steps=steps)      # optional outputs off
                assert torch.equal(lean["z_sorted"], zs) and torch.equal(lean["rgb"], c)
        assert torch.equal(torch.gather(torch.cat([z, zf], -1), -1, perm.long()), zs)
    torch.manual_seed(4)
    net = ana.NeRF(freqs_dir=0, use_view=False).to(dev)
    rays = torch.from_numpy(golden("frame")["rays_body"]).to(dev)
    for K in (3, 7, 64):
        steps = ana.VolumeRenderer(n_coarse=K)._table(dev, "steps", K)
        z = ops.sample_coarse(rays, steps).view(2, -1, K)
        for mode in ("f32", "bf16"):
            pack, mode_id = net.weight_pack(mode)
            assert torch.equal(ops.mlp_forward_rays(pack, mode_id, rays, z), ops.mlp_forward_rays_steps(pack, mode_id, rays, steps))
    # the renderer: lean schedule == general schedule, with and without the warp
    g = golden("render_cfg3_warp_gain")
    for warp in (False, True):
        m = seeded_model(smpl_table, g["seed"], warp, g["gain"], g["shift"], device=dev)
        with torch.no_grad():
            m.set_body_model(_to(tdict(g), dev), _templ(dev))
            rb = m.convert_to_body_model_space(torch.from_numpy(g["rays_world"]).to(dev))
            m.clac_ober2cano_transform()
            for Kc, Kf in ((64, 64), (64, 32), (32, 128)):
                vr = ana.VolumeRenderer(n_coarse=Kc, n_fine=Kf)
                a = vr(m, rb)
                vr.fuse_coarse_pass = False
                b = vr(m, rb)
                assert set(a) == set(b)
                for k in a:
                    assert torch.equal(a[k], b[k]), (k, warp, Kc, Kf)


# ----------------------------------------------------------------------------- a6-a15 end to end
CASES = ["cfg2_nowarp", "cfg2_nowarp_gain", "cfg3_warp_gain", "cfg1_coarse32_warp", "yaml_64_32_warp"]


@pytest.mark.parametrize("case", CASES)
def test_render_matches_reference(dev, smpl_table, case):
    """The five 64..144-ray reference renders: every ray within 1e-4 of the reference, or accounted for by a named
    discontinuity of the reference's own path (tests/accounting.py) — no percentage gate."""
    import anim_nerf_amd as ana
    from accounting import account_for_rays, render_stages
    from anim_nerf_amd import synthetic as syn
    g = golden("render_" + case)
    m = seeded_model(smpl_table, g["seed"], g["use_unpose"], g["gain"], g["shift"], device=dev, mlp_mode="f32")
    vr = ana.VolumeRenderer(n_coarse=int(g["n_coarse"]), n_fine=int(g["n_fine"]), white_bkgd=True)
    rays_w = torch.from_numpy(g["rays_world"])
    pose, templ = tdict(g), {k: torch.from_numpy(v) for k, v in syn.template_pose_params().items()}
    out = ana.batched_inference(vr, m, rays_w.to(dev), _to(pose, dev), _templ(dev), chunk=50)
    stages = render_stages(m, vr, rays_w, pose, templ)
    keys = ["rgbs", "alphas", "depths"] + (["rgbs_fine", "alphas_fine", "depths_fine"] if int(g["n_fine"]) else [])
    for k in keys:                                               # the chunk loop == one pass, bit for bit
        assert out[k].shape == g[k].shape and torch.equal(out[k], stages["out"][k]), k
    assert torch.equal(stages["zc"].cpu(), torch.from_numpy(g["z_coarse"]))
    torch.testing.assert_close(stages["rays_b"].cpu(), torch.from_numpy(g["rays_body"]), rtol=1e-5, atol=5e-6)
    stats = account_for_rays(m, vr, oracle_table(smpl_table), rays_w, pose, templ, g, stages=stages,
                             z_fine_ref=g.get("z_fine"), label=case)
    if not bool(g["use_unpose"]):
        # no warp, no discontinuity in the coarse pass: per-sample weights too
        w_ref = torch.from_numpy(g["weights"])[0]
        assert ((stages["w_c"].cpu().view_as(w_ref) - w_ref).abs() <= 2e-6 + RTOL * w_ref.abs()).all()
        if case == "cfg2_nowarp":
            assert stats["outside"] == 0, "literal initialisation, no warp: every ray within 1e-4"


@pytest.mark.parametrize("case", ["cfg2_nowarp_gain_4k", "cfg3_warp_gain_4k"])
def test_every_out_of_tolerance_ray_is_accounted_for(dev, smpl_table, case):
    """4,096-ray reference renders (64 + 64 samples, sigma gain 3000).  North-star tolerance: 1e-4 relative on every
    rendered value; every ray outside it is re-rendered by the ORACLE with the HIP path's own sorted depths, validity bits
    and (last) canonical points injected and must then agree within 1e-4 — 100 % of them — and must show its cause
    (tests/accounting.py)."""
    import anim_nerf_amd as ana
    from accounting import account_for_rays, render_stages
    from test_oracle_golden import big_case_inputs
    from anim_nerf_amd import synthetic as syn
    g = golden("render_" + case)
    warp = bool(g["use_unpose"])
    m = seeded_model(smpl_table, g["seed"], warp, g["gain"], g["shift"], device=dev, mlp_mode="f32")
    rays_w = big_case_inputs(g)
    pose, templ = tdict(g), {k: torch.from_numpy(v) for k, v in syn.template_pose_params().items()}
    vr = ana.VolumeRenderer(n_coarse=64, n_fine=64, white_bkgd=True)
    with torch.no_grad():
        out = ana.batched_inference(vr, m, rays_w.to(dev), _to(pose, dev), _templ(dev), chunk=4096)
    stages = render_stages(m, vr, rays_w, pose, templ)
    for k, v in stages["out"].items():
        assert torch.equal(out[k], v), k
    account_for_rays(m, vr, oracle_table(smpl_table), rays_w, pose, templ, g, stages=stages, z_fine_ref=g["z_fine"], label=case)


def test_warp_on_at_literal_init_every_ray_within_1e_4(dev, smpl_table):
    """configs[2] at the reference's LITERAL initialisation (no sigma gain; fixture render_cfg3_warp_init_1k.npz = the
    reference on 1,024 rays, 64 + 64 samples, animated pose): without the gain's amplification the warp is held to the
    north-star tolerance on EVERY ray — sampled depths, per-sample coarse weights, and the six rendered tensors.  The image
    is faint at this initialisation (opacity <= 0.04, colour within 0.02 of the white background), so colour and depth
    are compared as what they accumulate (1 - rgb, far' - depth): 1e-4 of THAT, not of a number next to 1.
    The only rays excused are those where a validity bit (distance to the 4 neighbours vs dis_threshold, anim_nerf.py:183)
    differs from the oracle's at the very same depth; they are counted and bounded."""
    import anim_nerf_amd as ana
    from test_oracle_golden import big_case_inputs
    from anim_nerf_amd import synthetic as syn
    g = golden("render_cfg3_warp_init_1k")
    assert float(g["gain"]) == 1.0 and float(np.abs(g["shift"]).max()) == 0.0
    m = seeded_model(smpl_table, g["seed"], True, 1.0, (0.0, 0.0), device=dev, mlp_mode="f32")
    assert weights_checksum(m.nerf) == str(g["w_coarse"]) and weights_checksum(m.nerf_fine) == str(g["w_fine"])
    rays_w = big_case_inputs(g)
    pose, templ = tdict(g), {k: torch.from_numpy(v) for k, v in syn.template_pose_params().items()}
    vr = ana.VolumeRenderer(n_coarse=64, n_fine=64, white_bkgd=True)
    with torch.no_grad():
        out = ana.batched_inference(vr, m, rays_w.to(dev), _to(pose, dev), _templ(dev), chunk=1024)
        m.set_body_model(_to(pose, dev), _templ(dev))
        rays_b = m.convert_to_body_model_space(rays_w.to(dev))
        m.clac_ober2cano_transform()
        zc = vr.sample_coarse(rays_b)
        w_c = vr._shade(m, rays_b, zc, True, 0.0, True)[0]
        zs = vr.sample_fine_sorted(zc, w_c)
        valid_c = m.warped_points(rays=rays_b, z=zc)[:, 3].view(1, -1, 64).cpu()
        valid_f = m.warped_points(rays=rays_b, z=zs)[:, 3].view(1, -1, 128).cpu()
    R = rays_w.shape[1]
    # the oracle's validity bits at the HIP path's own depths
    tbl = oracle_table(smpl_table)
    st = orc.frame_state(tbl, pose, templ)
    st, rays_o = orc.to_root_frame(st, rays_w)
    st["ober2cano"] = orc.observation_to_canonical(st)
    rb = rays_b.cpu()
    from accounting import neighbour_discontinuity
    flip = torch.zeros(R, dtype=torch.bool)
    for z_, v_hip in ((zc.cpu(), valid_c), (zs.cpu(), valid_f)):
        xyz = (rb[..., None, :3] + z_[..., None] * rb[..., None, 3:6]).reshape(1, -1, 3)
        _, valid_o, dbg = orc.warp_to_canonical(xyz, st["verts"], tbl["lbs_weights"], st["ober2cano"], 0.2, chunk=4096)
        flipped = valid_o.view(1, R, -1) != v_hip
        excuse = ((dbg["blended"][..., 0] - 0.2).abs() <= 1e-5) | neighbour_discontinuity(tbl["lbs_weights"], dbg["dist"], dbg["idx"])
        assert (flipped <= excuse.view(1, R, -1)).all(), "a validity bit differs from the oracle's away from the threshold"
        flip |= flipped.any(-1)[0]
    keep = ~flip
    print(f"\nliteral init: {int(flip.sum())} of {R} rays with a validity bit that is not the oracle's (each within rounding of "
          f"dis_threshold); {int((valid_f.sum(-1) > 0).sum())} rays touch the body")
    assert (valid_f.sum(-1) > 0).float().mean() > 0.2

    ref = {k: torch.from_numpy(g[k]) for k in ("rgbs", "alphas", "depths", "rgbs_fine", "alphas_fine", "depths_fine", "z_fine", "weights")}
    far = rb[..., 7:8]

    def held(name, a, b, atol):
        err = (a - b).abs()[:, keep]
        tol = (atol + RTOL * b.abs())[:, keep]
        assert (err <= tol).all(), (name, (err / tol).max().item(), int((err > tol).sum()))
    # sampled depths: the importance sampler sees no near-empty pdf at this initialisation, so no bin flips either
    zf_ref = torch.sort(torch.cat([zc.cpu(), ref["z_fine"]], -1), -1).values
    held("sorted depths", zs.cpu(), zf_ref, 2e-6)
    # per-sample coarse weights: alpha = 1 - exp(-delta sigma) with delta sigma ~ 1e-3 carries the rounding of exp() next
    # to 1 (1 ulp = 6e-8) in absolute terms
    held("weights", w_c.view(1, R, 64).cpu(), ref["weights"], 1.5e-7)
    got = {k: v.cpu() for k, v in out.items()}
    for tag in ("", "_fine"):
        held("alphas" + tag, got["alphas" + tag], ref["alphas" + tag], 2e-7)
        held("1 - rgbs" + tag, 1.0 - got["rgbs" + tag], 1.0 - ref["rgbs" + tag], 2e-7)
        held("far - depths" + tag, far - got["depths" + tag], far - ref["depths" + tag], 1e-6)
        for k in ("rgbs", "alphas", "depths"):                       # and the plain north-star statement, every kept ray
            held(k + tag, got[k + tag], ref[k + tag], 1e-6)


# BASELINE.json's metric is "rays/sec ...; PSNR vs ref" and the number bench.py reports is the bf16 mode's: the floor of that
# PSNR, against the REFERENCE's own outputs (the fixtures), per fixture.  (floor dB, max |rgb error| of any ray, max |alpha
# error|) = what the mode measured in round 5 (printed by the test) minus a margin of ~3 dB / x2: a regression of the bf16
# arithmetic (a rounding moved in front of an accumulation, an encoding octave lost) costs 6 dB and more.  Why bf16 cannot
# meet 1e-4: 8 bits of mantissa on every activation of an 8-layer network under a sigma gain of 3000.
BF16_FLOORS = {
    # measured (round 5):       rgb 71.8 / 61.7 dB (coarse / fine), opacity 65.9 / 55.7 dB, max |err| rgb 0.016, opacity 0.031
    "cfg2_nowarp_gain_4k": (52.0, 0.035, 0.065),
    # warp on:                  rgb 56.4 / 49.3 dB, opacity 50.3 / 43.1 dB, max |err| rgb 0.046, opacity 0.089
    "cfg3_warp_gain_4k": (40.0, 0.10, 0.18),
    # literal initialisation:   rgb 119 / 113 dB, opacity 113 / 107 dB, max |err| 2e-5 / 4e-5 (every ray inside 1e-4)
    "cfg3_warp_init_1k": (103.0, 1e-4, 1e-4),
}


def psnr_db(a, b):
    """models/evaluator.py:16-25 with data_range = 1 (colours and opacities live in [0, 1])"""
    return -10.0 * float(torch.log10(torch.mean((a.double() - b.double()) ** 2).clamp_min(1e-30)))


@pytest.mark.parametrize("case", list(BF16_FLOORS))
def test_timed_mode_bf16_psnr_against_the_reference(dev, smpl_table, case):
    """The mode bench.py times (bf16 MFMA) rendered on the reference's fixtures — no warp and warp on at sigma gain 3000
    (4,096 rays each) and warp on at the literal initialisation (1,024 rays), 64 + 64 samples — and held to the reference's
    OWN outputs: PSNR of the fine and the coarse image and of the opacity >= the floor, and every single ray inside a
    maximum absolute error (a mode that is right on average and wrong on a few rays fails the second gate)."""
    import anim_nerf_amd as ana
    from test_oracle_golden import big_case_inputs
    g = golden("render_" + case)
    floor_db, max_rgb, max_alpha = BF16_FLOORS[case]
    m = seeded_model(smpl_table, g["seed"], bool(g["use_unpose"]), g["gain"], g["shift"], device=dev, mlp_mode="bf16")
    rays_w = big_case_inputs(g)
    vr = ana.VolumeRenderer(n_coarse=64, n_fine=64, white_bkgd=True)
    with torch.no_grad():
        out = ana.batched_inference(vr, m, rays_w.to(dev), _to(tdict(g), dev), _templ(dev), chunk=4096)
    got = {k: v.cpu() for k, v in out.items()}
    report = {}
    for tag in ("", "_fine"):
        rgb_ref, a_ref = torch.from_numpy(g["rgbs" + tag]), torch.from_numpy(g["alphas" + tag])
        report["psnr_rgb" + tag] = psnr_db(got["rgbs" + tag], rgb_ref)
        report["psnr_alpha" + tag] = psnr_db(got["alphas" + tag], a_ref)
        report["max_rgb" + tag] = float((got["rgbs" + tag] - rgb_ref).abs().max())
        report["max_alpha" + tag] = float((got["alphas" + tag] - a_ref).abs().max())
        report["within_1e-4" + tag] = float(((got["rgbs" + tag] - rgb_ref).abs() <= 1e-6 + RTOL * rgb_ref.abs()).all(-1).float().mean())
    print(f"\nbf16 vs the reference [{case}]: " + ", ".join(f"{k} {v:.4g}" for k, v in report.items()))
    for tag in ("", "_fine"):
        assert report["psnr_rgb" + tag] >= floor_db and report["psnr_alpha" + tag] >= floor_db, (case, report)
        assert report["max_rgb" + tag] <= max_rgb and report["max_alpha" + tag] <= max_alpha, (case, report)


def test_jittered_coarse_depths_and_dead_twin_rays(dev):
    """anr_sample_coarse with t_rand (training jitter, models/volume_rendering.py:48-54) against the reference's output
    under the same uniforms; rays.get_ray_directions / get_rays (utils/ray_utils.py:74-121) against the reference's."""
    import anim_nerf_amd as ana
    g = golden("sampling_twins")
    rays = torch.from_numpy(g["rays"])
    for kc in (64, 32, 7):
        _, perturb, seed = g[f"cfg_{kc}"]
        torch.manual_seed(int(seed))
        t_rand = float(perturb) * torch.rand(*rays.shape[:2], kc)
        vr = ana.VolumeRenderer(n_coarse=kc)
        z = ana.ops.sample_coarse(rays.to(dev), vr._table(dev, "steps", kc), t_rand.view(-1, kc).to(dev))
        assert torch.equal(z.view(2, -1, kc).cpu(), torch.from_numpy(g[f"z_{kc}"])), kc
        # the module draws its own uniforms: every depth stays inside its stratum and the row stays sorted
        zj = vr.sample_coarse(rays.to(dev), perturb=1.0).cpu()
        z0 = orc.coarse_depths(rays, kc)
        mids = .5 * (z0[..., 1:] + z0[..., :-1])
        lo, hi = torch.cat([z0[..., :1], mids], -1), torch.cat([mids, z0[..., -1:]], -1)
        assert (zj >= lo).all() and (zj <= hi).all() and (zj[..., 1:] >= zj[..., :-1]).all()
    H, W, focal = int(g["twin_H"]), int(g["twin_W"]), float(g["twin_focal"])
    d = ana.rays.get_ray_directions(H, W, focal, device=dev)
    torch.testing.assert_close(d.cpu(), torch.from_numpy(g["twin_dirs"]), rtol=1e-6, atol=1e-7)
    ro, rd = ana.rays.get_rays(d, torch.from_numpy(g["twin_c2w"]).to(dev))
    torch.testing.assert_close(rd.cpu(), torch.from_numpy(g["twin_rays_d"]), rtol=1e-6, atol=1e-6)
    assert torch.equal(ro.cpu(), torch.from_numpy(g["twin_rays_o"]))


def test_rays_that_miss_the_body_render_background(dev, smpl_table):
    """A whole call without a single near / valid sample (empty lists all the way: no live cell, no MLP tile):
    white background, alpha 0, depth = far', in inference and under autograd."""
    import anim_nerf_amd as ana
    g = golden("render_cfg3_warp_gain")
    m = seeded_model(smpl_table, g["seed"], True, g["gain"], g["shift"], device=dev)
    world = torch.from_numpy(g["rays_world"]).to(dev).clone()
    world[..., 3:6] = -world[..., 3:6]                          # look away from the body
    vr = ana.VolumeRenderer(n_coarse=16, n_fine=8)
    for grad in (False, True):
        with torch.set_grad_enabled(grad):
            m.set_body_model(_to(tdict(g), dev), _templ(dev))
            rays = m.convert_to_body_model_space(world)
            m.clac_ober2cano_transform()
            out = vr(m, rays)
        for k in ("rgbs", "rgbs_fine"):
            assert torch.equal(out[k], torch.ones_like(out[k])), k
        for k in ("alphas", "alphas_fine"):
            assert out[k].abs().max() == 0, k
        assert torch.equal(out["depths_fine"][..., 0], rays[..., 7])
        if grad:
            out["rgbs_fine"].sum().backward()                   # nothing to differentiate, nothing to crash on
            assert all(p.grad is None or p.grad.abs().max() == 0 for p in m.nerf_fine.parameters())


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_sparse_paths_are_bit_identical_on_random_frames(dev, smpl_table, seed):
    """Random poses, camera distances, image sizes and sample counts (two bodies per call): the renderer's sparse path
    (bounding-box classification, cell sort, dead cells, per-cell radii, MLP on valid samples only) against the dense
    one (exact search everywhere, MLP everywhere) — every output tensor bit for bit, bf16 and fp32 MLP."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    rng = np.random.RandomState(seed)
    hw = int(rng.choice([9, 16, 23]))
    kc, kf = int(rng.choice([8, 33, 64])), int(rng.choice([0, 16, 64]))
    m = seeded_model(smpl_table, 10 + seed, True, 3000.0, (100.0, 100.0), device=dev)   # sigma > 0 wherever valid
    pose_np = syn.animated_pose_params(seed=50 + seed, bs=2, pose_std=0.35, transl_z=float(rng.uniform(-4.5, -2.0)))
    pose = {k: torch.from_numpy(v).to(dev) for k, v in pose_np.items()}
    c2w, focal, cen = syn.pinhole_camera(hw, hw)
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
    rays = rays.repeat(2, 1, 1)
    vr = ana.VolumeRenderer(n_coarse=kc, n_fine=kf)
    for mode in ("f32", "bf16"):
        m.nerf.mlp_mode = m.nerf_fine.mlp_mode = mode
        outs = []
        for sparse in (True, False):
            m.skip_far_samples = m.skip_invalid_samples = sparse
            with torch.no_grad():
                outs.append(ana.batched_inference(vr, m, rays, pose, _templ(dev), chunk=int(rng.choice([37, 4096]))))
        for k in outs[0]:
            assert torch.equal(outs[0][k], outs[1][k]), (k, mode, hw, kc, kf)
        key = "alphas_fine" if kf else "alphas"
        assert outs[0][key].max() > 0.2, "the body must be in view"


def test_cell_sorted_search_with_two_bodies_is_bit_identical(dev, smpl_table):
    """Two bodies per call at 2^19+ samples per body — the size from which the warp sorts its near samples by cell and
    searches them a wavefront per 64 neighbours (per-body rows of the cell workspace, of the item hand-out's counters and
    of the index): sparse == dense bit for bit, and the two bodies (different poses) really differ."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    hw = 96                                                         # 9,216 rays x 64 (128) samples = 589,824 (1.18 M) per body
    m = seeded_model(smpl_table, 17, True, 3000.0, (100.0, 100.0), device=dev, mlp_mode="bf16")
    pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=57, bs=2, pose_std=0.35, transl_z=-2.6).items()}
    c2w, focal, cen = syn.pinhole_camera(hw, hw)
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8).repeat(2, 1, 1)
    vr = ana.VolumeRenderer(n_coarse=64, n_fine=64)
    outs = []
    for sparse in (True, False):
        m.skip_far_samples = m.skip_invalid_samples = sparse
        with torch.no_grad():
            outs.append(ana.batched_inference(vr, m, rays, pose, _templ(dev), chunk=1 << 14))
    m.skip_far_samples = m.skip_invalid_samples = True
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert outs[0]["alphas_fine"].max() > 0.2, "the bodies must be in view"
    assert not torch.equal(outs[0]["rgbs_fine"][0], outs[0]["rgbs_fine"][1])
    # round 6: the sparse path computes the coarse depths from the step table inside the classify pass and the fused coarse
    # pass (no depth array); with the array (anr_sample_coarse, as before): the same bits
    vr.coarse_depth_array = True
    with torch.no_grad():
        with_array = ana.batched_inference(vr, m, rays, pose, _templ(dev), chunk=1 << 14)
    for k in outs[0]:
        assert torch.equal(outs[0][k], with_array[k]), k


def test_repeated_renders_are_bit_identical_next_to_other_work(dev, smpl_table):
    """One frame (two bodies, the cell-sorted sparse path, bf16) rendered 24 times while a side stream keeps the GPU busy with
    matrix products: every output of every render equals the first render's bit for bit.  The renderer has no float atomics —
    list orders are fixed by sorts and ordered compaction — so any difference is a fault; round 5 found one of that kind in the
    training step (packed fp32 adds of the SLP vectoriser misbehaving next to other kernels), this is the
    renderer's guard."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    hw = 96
    m = seeded_model(smpl_table, 17, True, 3000.0, (100.0, 100.0), device=dev, mlp_mode="bf16")
    pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=57, bs=2, pose_std=0.35, transl_z=-2.6).items()}
    c2w, focal, cen = syn.pinhole_camera(hw, hw)
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8).repeat(2, 1, 1)
    vr = ana.VolumeRenderer(n_coarse=64, n_fine=64)
    side = torch.cuda.Stream()
    a = torch.randn(1024, 1024, device=dev)
    first = None
    for it in range(24):
        with torch.cuda.stream(side):
            for _ in range(8):
                a = torch.tanh(a @ a) * 0.5
        with torch.no_grad():
            out = ana.batched_inference(vr, m, rays, pose, _templ(dev), chunk=1 << 14)
        out = {k: v.clone() for k, v in out.items()}
        if first is None:
            first = out
            assert first["alphas_fine"].max() > 0.2, "the bodies must be in view"
            continue
        for k in first:
            assert torch.equal(out[k], first[k]), (k, it, float((out[k].float() - first[k].float()).abs().max()))
    torch.cuda.synchronize()


def test_disparity_sampling_and_depth_guided_samples(dev, smpl_table):
    """The two VolumeRenderer options no shipped config selects: lindisp=False (models/volume_rendering.py:45-46) and
    n_fine_depth > 0 (:99-111, :204-207)."""
    import anim_nerf_amd as ana
    g = golden("render_cfg3_warp_gain")
    m = seeded_model(smpl_table, g["seed"], True, g["gain"], g["shift"], device=dev)
    with torch.no_grad():
        m.set_body_model(_to(tdict(g), dev), _templ(dev))
        rays = m.convert_to_body_model_space(torch.from_numpy(g["rays_world"]).to(dev))
        m.clac_ober2cano_transform()
        vr = ana.VolumeRenderer(n_coarse=16, n_fine=8, lindisp=False)
        z = vr.sample_coarse(rays)
        s = torch.linspace(0, 1 - 1.0 / 16, 16)
        r = rays.cpu()
        want = 1 / (1 / r[..., 6:7] * (1 - s) + 1 / r[..., 7:8] * s)
        torch.testing.assert_close(z.cpu(), want, rtol=1e-6, atol=1e-6)
        assert (z[..., 1:] > z[..., :-1]).all()
        out = vr(m, rays)
        gen = vr(lambda xyz, viewdir, use_fine=False: m(xyz, viewdir, use_fine=use_fine), rays)
        for k in out:
            torch.testing.assert_close(gen[k], out[k], rtol=2e-3, atol=2e-4)
        # depth-guided samples: Kc + Kf + Kfd sorted depths inside [near', far'], reproducible under a seed
        vd = ana.VolumeRenderer(n_coarse=16, n_fine=8, n_fine_depth=4, depth_std=0.02)
        torch.manual_seed(1)
        a = vd(m, rays)
        torch.manual_seed(1)
        b = vd(m, rays)
        for k in a:
            assert torch.equal(a[k], b[k]), k
        assert set(a) == set(out) and a["rgbs_fine"].shape == out["rgbs_fine"].shape
        zd = vd.sample_fine_depth(rays, a["depths"])
        assert zd.shape[-1] == 4 and (zd >= rays[..., 6:7]).all() and (zd <= rays[..., 7:8]).all()
        assert a["alphas_fine"].max() > 0.2


def test_share_fine_uses_one_network_and_returns_the_fine_triple(dev, smpl_table):
    """share_fine=True (models/anim_nerf.py:90-95, models/volume_rendering.py:219-224): nerf_fine IS nerf, and the renderer
    returns only the fine pass under the plain keys."""
    import anim_nerf_amd as ana
    g = golden("render_cfg3_warp_gain")
    torch.manual_seed(int(g["seed"]))
    m = ana.AnimNeRF(body_model_table=smpl_table, freqs_dir=0, use_view=False, use_unpose=True, use_fine=True,
                     share_fine=True).eval().to(dev)
    assert m.nerf_fine is m.nerf
    with torch.no_grad():
        m.nerf.sigma.weight.mul_(float(g["gain"]))
        m.nerf.sigma.bias.mul_(float(g["gain"])).add_(float(np.atleast_1d(g["shift"])[0]))
        m.set_body_model(_to(tdict(g), dev), _templ(dev))
        rays = m.convert_to_body_model_space(torch.from_numpy(g["rays_world"]).to(dev))
        m.clac_ober2cano_transform()
        shared = ana.VolumeRenderer(n_coarse=16, n_fine=8, share_fine=True)(m, rays)
        both = ana.VolumeRenderer(n_coarse=16, n_fine=8, share_fine=False)(m, rays)
    assert set(shared) == {"rgbs", "alphas", "depths"}
    for k in shared:
        assert torch.equal(shared[k], both[k + "_fine"]), k


def test_generic_model_path_equals_fused_path(dev, smpl_table):
    """VolumeRenderer.forward(model=<any callable>) hands materialised xyz to the model, as the reference does."""
    import anim_nerf_amd as ana
    g = golden("render_cfg3_warp_gain")
    m = seeded_model(smpl_table, g["seed"], True, g["gain"], g["shift"], device=dev)
    m.set_body_model(_to(tdict(g), dev), _templ(dev))
    rays = m.convert_to_body_model_space(torch.from_numpy(g["rays_world"]).to(dev))
    m.clac_ober2cano_transform()
    vr = ana.VolumeRenderer(n_coarse=64, n_fine=64)
    fused = vr(m, rays)
    # skipping the neighbour search for provably-invalid samples must not change a single bit of the render
    m.skip_far_samples = False
    exact_everywhere = vr(m, rays)
    m.skip_far_samples = True
    for k in fused:
        assert torch.equal(fused[k], exact_everywhere[k]), k
    # ... and neither must running the MLP on the valid samples only (the others composite with weight exactly 0)
    m.skip_invalid_samples = False
    dense = vr(m, rays)
    m.skip_invalid_samples = True
    for k in fused:
        assert torch.equal(fused[k], dense[k]), k
    generic = vr(lambda xyz, viewdir, use_fine=False: m(xyz, viewdir, use_fine=use_fine), rays)
    for k in fused:                 # same kernels; only x = o + z d is rounded differently (see conditioning note)
        torch.testing.assert_close(generic[k], fused[k], rtol=2e-3, atol=2e-4)


def test_mlp_from_rays_equals_points_then_mlp(dev, smpl_table):
    """anr_mlp_forward_rays (points generated inside the MLP kernel) = anr_points_from_rays + anr_mlp_forward, bit for bit."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import ops
    torch.manual_seed(4)
    net = ana.NeRF(freqs_dir=0, use_view=False).to(dev)
    rays = torch.from_numpy(golden("frame")["rays_body"]).to(dev)              # [2, R, 8]
    for K in (1, 7, 64):
        z = ana.VolumeRenderer(n_coarse=max(K, 3)).sample_coarse(rays)[..., :K].contiguous()
        for mode in ("f32", "bf16"):
            pack, mode_id = net.weight_pack(mode)
            a = ops.mlp_forward(pack, mode_id, ops.points_from_rays(rays, z))
            b = ops.mlp_forward_rays(pack, mode_id, rays, z)
            assert torch.equal(a, b), (K, mode)


def test_view_dependent_colour_head(dev, smpl_table):
    """NeRF(use_view=True) against the reference's output (tests/golden/mlp_view.npz).  Inference runs the whole network, the
    view-dependent colour head included, in the fused kernel (anr_mlp_forward_view); the path autograd takes — trunk / sigma /
    feature from the fused kernel + the head as library GEMMs — must give the same colours; get_sigma(only_sigma=False)
    returns the 256-wide feature."""
    import anim_nerf_amd as ana
    g = golden("mlp_view")
    torch.manual_seed(int(g["seed"]))
    net = ana.NeRF(freqs_xyz=10, freqs_dir=4, use_view=True, mlp_mode="f32").to(dev)
    xyz, vd = torch.from_numpy(g["xyz"]).to(dev), torch.from_numpy(g["viewdir"]).to(dev)
    with torch.no_grad():
        rgb, sig = net(xyz, vd)
        assert rel_err(rgb.cpu(), g["rgb"]) < RTOL
        s_ref = torch.from_numpy(g["sigma"])
        assert ((sig.cpu() - s_ref).abs() <= RTOL * s_ref.abs() + 1e-6).all()
        s2, feat = net.get_sigma(xyz)
        assert torch.equal(s2, sig)
        f_ref = torch.from_numpy(g["feature"])
        assert ((feat.cpu() - f_ref).abs() <= RTOL * f_ref.abs() + 2e-6).all()
        # the fused head against the head as framework ops on the kernel's feature (what training differentiates)
        pts = torch.cat([xyz.reshape(-1, 3), torch.ones_like(xyz.reshape(-1, 3)[:, :1])], -1)
        x = torch.cat([feat.reshape(-1, 256), net.encoding_dir(vd.reshape(-1, 3))], -1)
        rgb_ops = net.rgb(net.dir_encoding(x))
        assert (rgb.reshape(-1, 3) - rgb_ops).abs().max() < 2e-6
        # a direction per listed point: the compacted evaluation reads viewdir[index[i]]
        pts[::3, 3] = 0.0
        pack, mode_id = net.weight_pack("f32", view=True)
        sparse = ana.ops.mlp_forward_view(pack, mode_id, pts, vd.reshape(-1, 3).contiguous(), only_valid=True)
        keep = pts[:, 3] >= 1
        assert torch.equal(sparse[keep][:, :3], rgb.reshape(-1, 3)[keep]) and (sparse[~keep][:, 3] == -1e5).all()
        net.mlp_mode = "bf16"
        rgb16, _ = net(xyz, vd)
        assert (rgb16 - rgb).abs().max() < 2e-2
    # other octave counts of the direction encoding (the panel has room for 10)
    for freqs_dir in (0, 1, 10):
        torch.manual_seed(3)
        other = ana.NeRF(freqs_xyz=10, freqs_dir=freqs_dir, use_view=True, mlp_mode="f32").to(dev)
        with torch.no_grad():
            got, sg = other(xyz, vd)
            _, ft = other.get_sigma(xyz)
            want = other.rgb(other.dir_encoding(torch.cat([ft.reshape(-1, 256), other.encoding_dir(vd.reshape(-1, 3))], -1)))
        assert (got.reshape(-1, 3) - want).abs().max() < 2e-6, freqs_dir
    with pytest.raises(NotImplementedError):
        net.eval_points(torch.cat([xyz[0], torch.ones_like(xyz[0, :, :1])], -1))


def test_pre_embedded_twin_network(dev):
    """models/mlp.py's NeRF (models/mlp.py:226-297): forward(input_xyz[.., 63], input_dir, only_sigma) on the fused kernel
    with the encoder skipped, against the reference's outputs — an embedding, and 63 free channels."""
    import anim_nerf_amd as ana
    g = golden("mlp_twin")
    e_xyz = orc.fourier_encode(torch.from_numpy(g["xyz"]), 10).to(dev)
    e_dir = orc.fourier_encode(torch.from_numpy(g["viewdir"]), 4).to(dev)
    free = torch.from_numpy(g["free"]).to(dev)

    def close(a, ref, floor):
        ref = torch.from_numpy(ref)
        return ((a.cpu() - ref).abs() <= RTOL * ref.abs() + floor).all()
    for tag, dirs in (("view", 27), ("plain", 0)):
        torch.manual_seed(int(g["seed"]))
        net = ana.mlp.NeRF(in_channels_dir=dirs, mlp_mode="f32").to(dev)
        with pytest.raises(NotImplementedError):
            net(e_xyz, e_dir)                                  # inference-only: refuses under autograd
        with torch.no_grad():
            rgb, sig = net(e_xyz.view(2, -1, 63), e_dir.view(2, -1, 27) if dirs else None)
            assert rgb.shape == (2, 768, 3) and sig.shape == (2, 768, 1)
            assert close(rgb.view(-1, 3), g[f"{tag}_rgb"], 1e-6) and close(sig.view(-1, 1), g[f"{tag}_sigma"], 2e-6)
            rgb, sig = net(free, e_dir[:256] if dirs else None)
            assert close(rgb, g[f"{tag}_rgb_free"], 1e-6) and close(sig, g[f"{tag}_sigma_free"], 4e-6)
            so = net(e_xyz, only_sigma=True)
            assert so.shape == (1536, 1) and close(so, g[f"{tag}_only_sigma"], 2e-6)
            # the embedding of xyz through this entry = xyz through the encoder-fused entry (fp32: same polynomial? no -
            # torch's sin/cos vs the kernel's: within 1e-4)
            net.mlp_mode = "bf16"
            rgb16, sig16 = net(free, e_dir[:256] if dirs else None)
            assert (rgb16.cpu() - torch.from_numpy(g[f"{tag}_rgb_free"])).abs().max() < 3e-2


def test_view_dependent_render_matches_oracle(dev, smpl_table):
    """A whole coarse + fine render with use_view=True (generic renderer branch: ray directions per sample) against the
    oracle's renderer fed with the oracle's view-dependent field."""
    import anim_nerf_amd as ana
    g = golden("render_cfg3_warp_gain")
    torch.manual_seed(5)
    m = ana.AnimNeRF(body_model_table=smpl_table, freqs_dir=4, use_view=True, use_unpose=False, use_fine=True,
                     mlp_mode="f32").eval()
    with torch.no_grad():
        for net in (m.nerf, m.nerf_fine):
            net.sigma.weight.mul_(300.0)
            net.sigma.bias.mul_(300.0).add_(2.0)
    Pc, Pf = net_params(m.nerf), net_params(m.nerf_fine)
    m = m.to(dev)
    rays = torch.from_numpy(g["rays_world"])[:, :40]
    R = rays.shape[1]

    def field(xyz, use_fine):
        K = xyz.shape[1] // R
        vd = rays[..., None, 3:6].expand(-1, -1, K, -1).reshape(1, -1, 3)
        return orc.mlp_forward(Pf if use_fine else Pc, xyz, vd, use_view=True)
    ref = orc.render_rays(field, rays, 16, 8)
    with torch.no_grad():
        out = ana.VolumeRenderer(n_coarse=16, n_fine=8)(m, rays.to(dev))
    for k in ("rgbs", "alphas", "depths", "rgbs_fine", "alphas_fine", "depths_fine"):
        assert rel_err(out[k].cpu(), ref[k]) < RTOL, k
    assert out["alphas_fine"].max() > 0.5


def test_compact_valid_and_indexed_mlp(dev, smpl_table):
    """anr_compact_valid lists exactly the samples with valid >= 1; anr_mlp_forward_indexed gives those the bits the
    dense kernel gives them and leaves (0,0,0,-1e5) elsewhere (query_canonical_space_inside, models/anim_nerf.py:245-290)."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import ops
    torch.manual_seed(3)
    net = ana.NeRF(freqs_dir=0, use_view=False).to(dev)
    for n, frac in ((1, 1.0), (1, 0.0), (1023, 0.07), (70001, 0.07), (70001, 0.0), (4096, 1.0), (300000, 0.5)):
        g = torch.Generator().manual_seed(n)
        xyz = torch.rand(n, 3, generator=g) * 2 - 1
        valid = (torch.rand(n, 1, generator=g) < frac).float()
        pts = torch.cat([xyz, valid], 1).to(dev)
        fill = torch.full((n, 4), 7.0, device=dev)
        index, count = ops.compact_valid(pts, fill=fill)
        cnt = int(count.item())
        want = torch.nonzero(valid[:, 0] >= 1)[:, 0]
        assert cnt == want.numel()
        assert torch.equal(index[:cnt].long().sort().values.cpu(), want)
        inv = (valid[:, 0] < 1).to(dev)
        assert torch.equal(fill[inv], torch.tensor([0.0, 0.0, 0.0, -1e5], device=dev).expand(int(inv.sum()), 4))
        assert (fill[~inv] == 7.0).all()
        for mode in ("f32", "bf16"):
            pack, mode_id = net.weight_pack(mode)
            dense = ops.mlp_forward(pack, mode_id, pts)
            sparse = ops.mlp_forward(pack, mode_id, pts, only_valid=True)
            assert torch.equal(sparse[~inv], dense[~inv]), (n, frac, mode)
            assert torch.equal(sparse[inv], fill[inv])
            assert torch.equal(dense[inv][:, 3], fill[inv][:, 3])
            s_dense = ops.mlp_forward(pack, mode_id, pts, sigma_only=True)
            s_sparse = ops.mlp_forward(pack, mode_id, pts, sigma_only=True, only_valid=True)
            assert torch.equal(s_dense, s_sparse)


def test_query_inside(dev, smpl_table):
    """AnimNeRF(query_inside=True): sigma as query_inside=False, rgb = 0 where the warp is invalid."""
    g = golden("render_cfg3_warp_gain")
    outs = []
    for qi in (False, True):
        m = seeded_model(smpl_table, g["seed"], True, g["gain"], g["shift"], device=dev)
        m.query_inside = qi
        with torch.no_grad():
            m.set_body_model(_to(tdict(g), dev), _templ(dev))
            rays = m.convert_to_body_model_space(torch.from_numpy(g["rays_world"]).to(dev))
            m.clac_ober2cano_transform()
            z = torch.linspace(2.0, 4.0, 48, device=dev).expand(1, rays.shape[1], 48)
            xyz = (rays[..., None, :3] + z[..., None] * rays[..., None, 3:6]).reshape(1, -1, 3)
            outs.append(m(xyz))
    (rgb0, sig0), (rgb1, sig1) = outs
    assert torch.equal(sig0, sig1)
    inv = sig0[..., 0] == -1e5
    assert 0.02 < (~inv).float().mean() < 0.9
    assert torch.equal(rgb1[~inv], rgb0[~inv]) and (rgb1[inv] == 0).all()


# ----------------------------------------------------------------------------- full BASELINE size
def test_full_frame_properties(dev, smpl_table):
    """1024 x 1024, 64 + 64 (BASELINE config 2) through size-independent properties:
    determinism, chunk invariance, sortedness (inside the kernel tests), alpha range, fp32-vs-bf16 PSNR."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    g = golden("render_cfg2_nowarp_gain")
    m = seeded_model(smpl_table, g["seed"], False, g["gain"], g["shift"], device=dev, mlp_mode="bf16")
    H = W = 1024
    c2w, focal, cen = syn.pinhole_camera(H, W)
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), H, W, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
    pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.static_pose_params().items()}
    vr = ana.VolumeRenderer(n_coarse=64, n_fine=64)
    a = ana.batched_inference(vr, m, rays, pose, _templ(dev), chunk=1 << 18)
    b = ana.batched_inference(vr, m, rays, pose, _templ(dev), chunk=1 << 18)
    c = ana.batched_inference(vr, m, rays, pose, _templ(dev), chunk=100003)
    for k in a:
        assert torch.equal(a[k], b[k]), "render must be deterministic"
        assert torch.equal(a[k], c[k]), "render must not depend on the chunking"
        assert torch.isfinite(a[k]).all()
    assert a["alphas_fine"].min() >= 0 and a["alphas_fine"].max() <= 1 + 1e-5
    assert a["rgbs_fine"].min() >= 0 and a["rgbs_fine"].max() <= 1 + 1e-5
    # centre 256 x 256 crop in fp32 (parity mode) vs bf16: PSNR
    idx = (torch.arange(384, 640)[:, None] * W + torch.arange(384, 640)[None]).reshape(-1).to(dev)
    m.nerf.mlp_mode = m.nerf_fine.mlp_mode = "f32"
    f = ana.batched_inference(vr, m, rays[:, idx].contiguous(), pose, _templ(dev), chunk=1 << 16)
    psnr = orc.psnr(a["rgbs_fine"][:, idx].cpu(), f["rgbs_fine"].cpu())
    assert psnr > 35.0, psnr


def test_one_pass_ray_march_equals_the_staged_path(dev, smpl_table):
    """The one-pass ray-march kernel (anr_ray_march, csrc/ray_march.hip: stratified samples, point generation, encoding, coarse
    network, compositing, importance sampling + merge, fine network, compositing in ONE launch) against the staged launches
    (anr_mlp_forward_rays_steps + anr_composite_sample + anr_mlp_forward_rays + anr_composite) on the no-warp model of the
    reference fixture: every output tensor BIT FOR BIT, fp32 and bf16, ragged ray counts (1, 3, 4, 5, 259 rays: tail groups,
    fewer groups than workgroups) and a 300 x 300 image (more groups than workgroups: the persistent loop, the ring wrapping
    between the two networks' packs), chunked and not.  The staged path's gates against the reference (every ray within 1e-4 or
    accounted for: test_render_matches_reference, test_every_out_of_tolerance_ray_is_accounted_for) are thereby the one-pass
    kernel's.  Shapes other than 64 + 64 are refused by the entry point, not mis-rendered."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import ops, synthetic as syn
    g = golden("render_cfg2_nowarp_gain")
    pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.static_pose_params().items()}
    c2w, focal, cen = syn.pinhole_camera(300, 300)
    big = ana.gen_rays(torch.from_numpy(c2w).to(dev), 300, 300, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
    gen = torch.Generator().manual_seed(5)
    for mode in ("f32", "bf16"):
        m = seeded_model(smpl_table, g["seed"], False, g["gain"], g["shift"], device=dev, mlp_mode=mode)
        staged, fused = ana.VolumeRenderer(n_coarse=64, n_fine=64), ana.VolumeRenderer(n_coarse=64, n_fine=64)
        staged.one_pass, fused.one_pass = False, True
        sets = [big[:, torch.randperm(big.shape[1], generator=gen)[:n].to(dev)].contiguous() for n in (1, 3, 4, 5, 259)]
        sets.append(big if mode == "bf16" else big[:, :20000].contiguous())
        for rays in sets:
            with torch.no_grad():
                ops.KERNEL_TIMING = []
                a = ana.batched_inference(fused, m, rays, pose, _templ(dev), chunk=1 << 20)
                names = {k[0] for k in ops.KERNEL_TIMING}
                ops.KERNEL_TIMING = None
                assert "ray_march" in names and "mlp_forward" not in names and "composite" not in names, names
                b = ana.batched_inference(staged, m, rays, pose, _templ(dev), chunk=1 << 20)
                c = ana.batched_inference(fused, m, rays, pose, _templ(dev), chunk=1001)
            assert set(a) == set(b) == {"rgbs", "alphas", "depths", "rgbs_fine", "alphas_fine", "depths_fine"}
            for k in a:
                assert torch.equal(a[k], b[k]), (k, mode, rays.shape[1])
                assert torch.equal(a[k], c[k]), (k, mode, rays.shape[1], "chunked")
        assert b["alphas_fine"].max() > 0.5, "the field must be visible"
    lib = ana._lib.load()
    z = torch.zeros(64, device=dev)
    rc = lib.anr_ray_march(z.data_ptr(), z.data_ptr(), 1, z.data_ptr(), 8, 4, z.data_ptr(), 64, z.data_ptr(), 32, 1, *([z.data_ptr()] * 6), None)
    assert rc < 0 and b"64 + 64" in lib.anr_last_error()


@pytest.mark.parametrize("seed", [0, 1])
def test_one_pass_ray_march_with_the_warp_equals_the_staged_path(dev, smpl_table, seed):
    """anr_ray_march_warp: the one-pass kernel with the inverse-LBS / exact 4-NN warp INSIDE the pass (every sample warped where
    it is generated — index read from global memory with the search and blend routines of warp_core.h —, every sample through
    the networks, sigma masked where invalid) against the staged renderer (classify / cells / cell-sorted search / valid list /
    MLP on the valid samples / masked compositors): two bodies with different poses per call, ray counts that are no multiple of
    the kernel's four-ray groups (a group straddles the two bodies), random camera distance — every output tensor BIT FOR BIT in
    fp32 and bf16.  The staged path's gates against the reference (accounting, tests above) are thereby this kernel's too."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import ops, synthetic as syn
    rng = np.random.RandomState(100 + seed)
    hw = int(rng.choice([17, 23]))
    m = seeded_model(smpl_table, 30 + seed, True, 3000.0, (100.0, 100.0), device=dev)
    pose_np = syn.animated_pose_params(seed=70 + seed, bs=2, pose_std=0.35, transl_z=float(rng.uniform(-4.0, -2.2)))
    pose = {k: torch.from_numpy(v).to(dev) for k, v in pose_np.items()}
    c2w, focal, cen = syn.pinhole_camera(hw, hw)
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), hw, hw, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8).repeat(2, 1, 1)
    staged, fused = ana.VolumeRenderer(n_coarse=64, n_fine=64), ana.VolumeRenderer(n_coarse=64, n_fine=64)
    staged.one_pass, fused.one_pass = False, True
    for mode in ("f32", "bf16"):
        m.nerf.mlp_mode = m.nerf_fine.mlp_mode = mode
        with torch.no_grad():
            ops.KERNEL_TIMING = []
            a = ana.batched_inference(fused, m, rays, pose, _templ(dev), chunk=1 << 20)
            names = {k[0] for k in ops.KERNEL_TIMING}
            ops.KERNEL_TIMING = None
            assert "ray_march_warp" in names and "warp_points" not in names and "mlp_forward" not in names, names
            b = ana.batched_inference(staged, m, rays, pose, _templ(dev), chunk=1 << 20)
            c = ana.batched_inference(fused, m, rays, pose, _templ(dev), chunk=101)
        for k in b:
            assert torch.equal(a[k], b[k]), (k, mode, hw)
            assert torch.equal(a[k], c[k]), (k, mode, hw, "chunked")
        assert b["alphas_fine"].max() > 0.2, "the bodies must be in view"
        assert not torch.equal(b["rgbs_fine"][0], b["rgbs_fine"][1])


def test_full_frame_properties_with_the_warp(dev, smpl_table):
    """1024 x 1024, 64 + 64, inverse-LBS / 4-NN warp on (BASELINE configs[2]) — the sparse machinery at its real size
    (64^3 cell grid, dead cells, 2^20-ray lists, validity bytes, coarse->fine reuse): determinism, chunk invariance,
    sparse == dense bit for bit on a 256 x 256 crop, K5 on the rays that miss the body, alpha range, bf16 vs fp32 PSNR."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    g = golden("render_cfg3_warp_gain")
    m = seeded_model(smpl_table, g["seed"], True, g["gain"], g["shift"], device=dev, mlp_mode="bf16")
    H = W = 1024
    c2w, focal, cen = syn.pinhole_camera(H, W)
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), H, W, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
    pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=100).items()}
    vr = ana.VolumeRenderer(n_coarse=64, n_fine=64)
    a = ana.batched_inference(vr, m, rays, pose, _templ(dev), chunk=1 << 20)
    b = ana.batched_inference(vr, m, rays, pose, _templ(dev), chunk=1 << 20)
    c = ana.batched_inference(vr, m, rays, pose, _templ(dev), chunk=100003)
    for k in a:
        assert torch.equal(a[k], b[k]), "render must be deterministic"
        assert torch.equal(a[k], c[k]), "render must not depend on the chunking"
        assert torch.isfinite(a[k]).all()
    af = a["alphas_fine"][0, :, 0]
    assert af.min() >= 0 and af.max() <= 1 + 1e-5 and a["rgbs_fine"].min() >= 0 and a["rgbs_fine"].max() <= 1 + 1e-5
    covered = (af > 0.5).float().mean().item()
    assert 0.03 < covered < 0.6, covered                       # a body in front of an empty background
    # K5: rays without a single valid sample are exactly white, alpha 0, depth = far'
    m.set_body_model(pose, _templ(dev))
    rays_b = m.convert_to_body_model_space(rays)
    m.clac_ober2cano_transform()
    empty = (a["alphas_fine"][0, :, 0] == 0) & (a["alphas"][0, :, 0] == 0)
    assert empty.float().mean() > 0.3
    assert torch.equal(a["rgbs_fine"][0, empty], torch.ones_like(a["rgbs_fine"][0, empty]))
    assert torch.equal(a["depths_fine"][0, empty, 0], rays_b[0, empty, 7])
    # sparse (cells, dead cells, valid-only MLP, lean schedule) == dense (exact search everywhere, MLP everywhere)
    idx = (torch.arange(384, 640)[:, None] * W + torch.arange(384, 640)[None]).reshape(-1).to(dev)
    crop = rays[:, idx].contiguous()
    sparse = ana.batched_inference(vr, m, crop, pose, _templ(dev), chunk=1 << 16)
    m.skip_far_samples = m.skip_invalid_samples = False
    dense = ana.batched_inference(vr, m, crop, pose, _templ(dev), chunk=1 << 14)
    m.skip_far_samples = m.skip_invalid_samples = True
    for k in sparse:
        assert torch.equal(sparse[k], dense[k]), k
        assert torch.equal(sparse[k], a[k][:, idx]), k         # ... and the crop rendered alone == the crop of the frame
    m.nerf.mlp_mode = m.nerf_fine.mlp_mode = "f32"
    f = ana.batched_inference(vr, m, crop, pose, _templ(dev), chunk=1 << 16)
    psnr = orc.psnr(sparse["rgbs_fine"].cpu(), f["rgbs_fine"].cpu())
    assert psnr > 30.0, psnr


# ----------------------------------------------------------------------------- BASELINE configs[4]: sigma grid
def test_sigma_grid_matches_reference_loop(dev, smpl_table):
    """extract_mesh.py:27-35,49-61,152-158: create_grid + centre + chunked AnimNeRF.forward + relu, vs the sharded
    sigma-only fast path (device grid, no search for provably-empty voxels, MLP stops at the sigma row)."""
    import anim_nerf_amd as ana
    m = _warp_frame(dev, smpl_table)
    m.verts, m.ober2cano_transform = m.verts[:1].contiguous(), m.ober2cano_transform[:1].contiguous()
    N = 24
    rng = (-1.2, 1.2)
    lin = np.linspace(rng[0], rng[1], N)
    grid = np.stack(np.meshgrid(lin, lin, lin), -1).reshape(-1, 3)               # the reference's create_grid
    center = (m.verts.max(dim=1)[0] + m.verts.min(dim=1)[0]) / 2.
    points = torch.from_numpy(grid).unsqueeze(0).float().to(dev) + center
    ref = ana.sigma_grid_inference(m, points, chunk=5000)[0, :, 0]                  # exact everywhere, full MLP
    # oracle on a subset (CPU brute force), voxel by voxel: within 1e-4 or a named discontinuity
    from accounting import account_for_points
    sub = torch.arange(0, N ** 3, 7, device=dev)
    st = account_for_points(m, oracle_table(smpl_table), points[:, sub], ref[sub], use_fine=True, relu=True, label="sigma grid 24^3")
    assert st["valid"] > 20
    parts = [ana.sigma_grid(m, N, rng, rng, rng, chunk=3000, rank=r, world=3) for r in range(3)]
    assert [p[1] for p in parts] == [ana.shard_range(N ** 3, r, 3)[0] for r in range(3)]
    fast = torch.cat([p[0] for p in parts])
    assert fast.shape == ref.shape
    assert torch.equal(fast, ref), (fast - ref).abs().max()
    assert (fast > 0).any() and (fast == 0).float().mean() > 0.5                  # a body in mostly empty space


def test_drivers_novel_view_and_sigma_grid_export(dev, tmp_path):
    """The runnable drivers (novel_view.py:144-210, extract_mesh.py:142-173 minus marching cubes) on the seeded synthetic
    scene: view 0 of the orbit is the plain render, the PNGs decode to it, the exported volume is sigma_grid - threshold."""
    import struct
    import zlib
    import anim_nerf_amd as ana
    out = ana.drivers.main(["novel_view", "--synthetic", "--n_views", "3", "--img_wh", "40", "32", "--out", str(tmp_path / "nv"),
                            "--mlp_mode", "f32"])
    args = ana.drivers.parser().parse_args(["novel_view", "--synthetic", "--img_wh", "40", "32", "--out", "x", "--mlp_mode", "f32"])
    model, vr, rays, pose, templ = ana.drivers._synthetic_scene(args, dev)
    ref = ana.batched_inference(vr, model, rays.view(1, -1, 8), pose, templ, chunk=1 << 20)
    want = (ref["rgbs_fine"].view(32, 40, 3).clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy()

    def read_png(path):
        b = open(path, "rb").read()
        i, idat, hdr = 8, b"", None
        while i < len(b):
            n = struct.unpack(">I", b[i:i + 4])[0]
            tag, pay = b[i + 4:i + 8], b[i + 8:i + 8 + n]
            if tag == b"IHDR":
                hdr = struct.unpack(">IIBBBBB", pay)
            if tag == b"IDAT":
                idat += pay
            i += 12 + n
        w, h, _, colour = hdr[:4]
        c = {0: 1, 2: 3, 6: 4}[colour]
        raw = zlib.decompress(idat)
        return np.frombuffer(b"".join(raw[y * (1 + w * c) + 1:(y + 1) * (1 + w * c)] for y in range(h)), np.uint8).reshape(h, w, c)
    img0 = read_png(f"{out}/images/000000.png")
    assert img0.shape == (32, 40, 4) and np.array_equal(img0[..., :3], want)          # P_0 = identity
    assert not np.array_equal(read_png(f"{out}/images/000001.png")[..., :3], want)   # the orbit moves
    assert read_png(f"{out}/depths/000002.png").shape == (32, 40, 1)
    # novel_pose.py:118-176: the subject driven through a (seeded) motion sequence; frame 0 is the plain render of its parameters
    outp = ana.drivers.main(["novel_pose", "--synthetic", "--n_frames", "4", "--frame_skip", "2", "--img_wh", "40", "32",
                             "--out", str(tmp_path / "np"), "--mlp_mode", "f32"])
    a0, a1 = read_png(f"{outp}/images/000000.png"), read_png(f"{outp}/images/000001.png")
    assert a0.shape == (32, 40, 4) and not np.array_equal(a0, a1) and not os.path.exists(f"{outp}/images/000002.png")
    assert np.array_equal(read_png(f"{outp}/masks/000000.png")[..., 0], a0[..., 3]) and read_png(f"{outp}/depths/000001.png").shape == (32, 40, 1)
    assert (a0[..., 3] > 128).mean() > 0.02, "the body must be in the picture"
    out = ana.drivers.main(["extract_grid", "--synthetic", "--N_grid", "24", "--sigma_threshold", "5", "--out", str(tmp_path / "m"),
                            "--mlp_mode", "f32"])
    vol = np.load(f"{out}/sigma.npy")
    with torch.no_grad():
        model.set_body_model(pose, templ)
        model.convert_to_body_model_space(rays.view(1, -1, 8)[:, :1])
        model.clac_ober2cano_transform()
        sig, _ = ana.sigma_grid(model, 24)
    assert vol.shape == (24, 24, 24) and np.array_equal(vol.reshape(-1), sig.cpu().numpy() - 5.0)
    assert (vol > 0).any() and open(f"{out}/smpl.obj").readline().startswith("v ")
    # mesh.obj (extract_mesh.py:165-173): the level set of that volume, closed, every vertex on a grid edge that straddles the
    # threshold, placed in the world by the reference's rescale (/ N, x and y swapped) + the body's centre
    from test_mesh import check_closed_oriented_surface
    lines = open(f"{out}/mesh.obj").read().split("\n")
    mv = np.array([[float(x) for x in ln.split()[1:]] for ln in lines if ln.startswith("v ")])
    mf = np.array([[int(x) - 1 for x in ln.split()[1:]] for ln in lines if ln.startswith("f ")])
    assert mv.shape[0] > 0 and mf.min() == 0 and mf.max() == mv.shape[0] - 1
    check_closed_oriented_surface(mv, mf)
    crossings = sum(int(((np.take(vol, range(0, 23), a) > 0) != (np.take(vol, range(1, 24), a) > 0)).sum()) for a in range(3))
    assert mv.shape[0] == crossings
    center = np.load(f"{out}/center.npy")
    idx = (mv - center)[:, [1, 0, 2]]                                      # undo the swap ...
    idx = (idx + 1.2) / 2.4 * 24                                           # ... and the rescale: index coordinates again
    assert idx.min() >= 0 and idx.max() <= 23 and (np.abs(idx - np.round(idx)) < 1e-5).sum(1).min() >= 2     # on grid edges


# ----------------------------------------------------------------------------- a2: SMPL / LBS kernels
def test_smpl_kernels_match_oracle(dev, smpl_table):
    """anr_smpl_forward (3 launches) vs the oracle's restatement of smplx/lbs.py, bs = 3, all six outputs."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    bm = ana.SMPL(data_struct=smpl_table).to(dev)
    pose = {k: torch.from_numpy(v) for k, v in syn.animated_pose_params(seed=9, bs=3, pose_std=0.4).items()}
    with torch.no_grad():
        o = bm(**{k: v.to(dev) for k, v in pose.items()})
    ref = orc.smpl_forward(oracle_table(smpl_table), **pose)
    for k in ("vertices", "joints", "joints_transform", "vertices_transform", "shape_offsets", "pose_offsets"):
        assert o[k].shape == ref[k].shape, k
        torch.testing.assert_close(o[k].cpu(), ref[k], rtol=1e-5, atol=3e-6, msg=lambda m: f"{k}: {m}")
    # the frame state is reproducible bit for bit (no float atomics in the joint regression)
    with torch.no_grad():
        for _ in range(3):
            again = bm(**{k: v.to(dev) for k, v in pose.items()})
            for k in ("vertices", "joints", "joints_transform", "vertices_transform"):
                assert torch.equal(again[k], o[k]), k
    # the autograd form (pose refinement) agrees with the kernels
    g = {k: v.to(dev).requires_grad_(True) for k, v in pose.items()}
    o2 = bm(**g)
    assert o2["vertices"].requires_grad
    torch.testing.assert_close(o2["vertices"].detach(), o["vertices"], rtol=1e-5, atol=3e-6)
    torch.testing.assert_close(o2["vertices_transform"].detach(), o["vertices_transform"], rtol=1e-5, atol=3e-6)


def test_unpose_view_matches_reference(dev, smpl_table):
    """AnimNeRF(use_view=True, unpose_view=True) — unpose() and forward() — against the reference's outputs
    (tests/golden/unpose_view.npz): canonical points, carried view directions, validity, rgb and sigma of both networks."""
    from anim_nerf_amd import synthetic as syn
    from test_oracle_golden import _unpose_view_model
    g = golden("unpose_view")
    m = _unpose_view_model(smpl_table, g, dev)
    pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=1, bs=2).items()}
    xyz, vd = torch.from_numpy(g["xyz"]).to(dev), torch.from_numpy(g["viewdir"]).to(dev)
    with torch.no_grad():
        m.set_body_model(pose, _templ(dev))
        m.convert_to_body_model_space(torch.from_numpy(g["rays_world"]).to(dev))
        m.clac_ober2cano_transform()
        xc, vc, valid = m.unpose(xyz, vd)
        # (the confidence threshold of the blend flips on fp32 rounding for a few samples in 10^4: models/anim_nerf.py:166)
        x_ref, v_ref = torch.from_numpy(g["xyz_c"]), torch.from_numpy(g["viewdir_c"])
        ok = ((xc.cpu() - x_ref).abs().max(-1).values <= 1e-5 + RTOL * x_ref.abs().max(-1).values)
        assert ok.float().mean() > 0.998
        assert ((vc.cpu() - v_ref).abs().max(-1).values[ok] <= 1e-5 + RTOL * v_ref.abs().max(-1).values[ok]).all()
        assert (valid.cpu() == torch.from_numpy(g["valid"]))[ok].all()
        for tag, fine in (("", False), ("_fine", True)):
            rgb, sigma = m(xyz, vd, use_fine=fine)
            r_ref, s_ref = torch.from_numpy(g["rgb" + tag]), torch.from_numpy(g["sigma" + tag])
            assert ((rgb.cpu() - r_ref).abs().max(-1).values[ok] <= 1e-5 + RTOL).all()
            assert ((sigma.cpu() - s_ref).abs()[ok] <= 1e-5 + RTOL * s_ref.abs()[ok]).all()


@pytest.mark.parametrize("k", [3, 6])
def test_other_neighbour_counts_match_reference(dev, smpl_table, k):
    """AnimNeRF(k_neigh=3 / 6): anr_knn_k (exhaustive exact search) + the blend as tensor ops against the reference's
    unpose() / forward() (tests/golden/kneigh.npz), the kernel's neighbours against a brute-force top-k, and a render
    through the general renderer branch against the oracle's."""
    from anim_nerf_amd import ops, synthetic as syn
    import anim_nerf_amd as ana
    from test_oracle_golden import _kneigh_model
    g = golden("kneigh")
    m = _kneigh_model(smpl_table, g, k, dev)
    pose = {kk: torch.from_numpy(v).to(dev) for kk, v in syn.animated_pose_params(seed=1, bs=2).items()}
    xyz = torch.from_numpy(g["xyz"]).to(dev)
    with torch.no_grad():
        m.set_body_model(pose, _templ(dev))
        rays_b = m.convert_to_body_model_space(torch.from_numpy(g["rays_world"]).to(dev))
        m.clac_ober2cano_transform()
        dist, idx = ops.knn_k(m.verts, xyz, k)
        d_all = torch.norm(xyz[:, :, None] - m.verts[:, None], dim=-1)
        d_ref, i_ref = d_all.topk(k, largest=False, dim=-1)
        torch.testing.assert_close(dist, d_ref, rtol=1e-6, atol=1e-6)
        assert ((idx == i_ref) | ((dist - d_ref).abs() <= 1e-6)).all() and (idx == i_ref).float().mean() > 0.999
        xc, _, valid = m.unpose(xyz)
        x_ref = torch.from_numpy(g[f"xyz_c_{k}"])
        ok = (xc.cpu() - x_ref).abs().max(-1).values <= 1e-5 + RTOL * x_ref.abs().max(-1).values
        assert ok.float().mean() > 0.998                      # confidence-threshold flips on fp32 rounding aside
        assert (valid.cpu() == torch.from_numpy(g[f"valid_{k}"]))[ok].all()
        rgb, sigma = m(xyz, None, use_fine=False)
        s_ref = torch.from_numpy(g[f"sigma_{k}"])
        assert ((rgb.cpu() - torch.from_numpy(g[f"rgb_{k}"])).abs().max(-1).values[ok] <= 1e-5 + RTOL).all()
        assert ((sigma.cpu() - s_ref).abs()[ok] <= 2e-4 + RTOL * s_ref.abs()[ok]).all()
        out = ana.VolumeRenderer(n_coarse=16, n_fine=8)(m, rays_b)
    tbl = oracle_table(smpl_table)
    st = orc.frame_state(tbl, {kk: v.cpu() for kk, v in pose.items()}, {kk: torch.from_numpy(v) for kk, v in syn.template_pose_params().items()})
    st, rays_o = orc.to_root_frame(st, torch.from_numpy(g["rays_world"]))
    st["ober2cano"] = orc.observation_to_canonical(st)
    Pc, Pf = net_params(m.nerf), net_params(m.nerf_fine)

    def field(p, use_fine):
        xc, valid, _ = orc.warp_to_canonical(p, st["verts"], tbl["lbs_weights"], st["ober2cano"], 0.2, k=k, chunk=1024)
        rgb, sig = orc.mlp_forward(Pf if use_fine else Pc, xc)
        return rgb, torch.where(valid < 1, torch.full_like(sig, -1e5), sig)
    ref = orc.render_rays(field, rays_o, 16, 8)
    bad = 0
    for key in ("rgbs", "alphas", "depths", "rgbs_fine", "alphas_fine", "depths_fine"):
        bad = bad | ((out[key].cpu() - ref[key]).abs() > 1e-5 + 1e-3 * ref[key].abs()).any(-1)
    assert bad.float().mean() <= 0.1, bad.float().mean()       # 16 rays per body: a flipped sample moves a whole ray


def test_marching_cubes_kernels(dev):
    """anr_mc_classify / anr_mc_emit (extract_mesh.py:165: mcubes.marching_cubes(-sigmas, 0.); PyMCubes is absent, so: the
    numpy cube-by-cube restatement of tests/test_mesh.py on small volumes — same vertices, same triangles — and properties at
    size: a closed, consistently oriented surface whose vertices lie on the level set)."""
    import anim_nerf_amd as ana
    from test_mesh import check_closed_oriented_surface, numpy_marching_cubes
    rng = np.random.default_rng(3)
    for shape in ((7, 9, 6), (12, 12, 12)):
        g = np.stack(np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij"), -1)
        f = (np.linalg.norm(g, axis=-1) - 0.6 + 0.3 * rng.standard_normal(shape)).astype(np.float32)      # ambiguous faces included
        v_ref, t_ref = numpy_marching_cubes(f)
        v, t = ana.mesh.marching_cubes(torch.from_numpy(f).to(dev), 0.0)
        v, t = v.cpu().numpy().astype(np.float64), t.cpu().numpy()
        assert v.shape == v_ref.shape and t.shape == t_ref.shape
        # same vertex set (the order differs: grid point, then axis) ...
        key = lambda a: np.lexsort(np.round(a * 4096).astype(np.int64).T[::-1])
        o, o_ref = key(v), key(v_ref)
        np.testing.assert_allclose(v[o], v_ref[o_ref], atol=2e-6)
        # ... and the same triangles on it, up to a rotation of their corners
        rank, rank_ref = np.empty_like(o), np.empty_like(o_ref)
        rank[o], rank_ref[o_ref] = np.arange(len(o)), np.arange(len(o_ref))
        canon = lambda tri: {tuple(np.roll(r, -int(np.argmin(r)))) for r in tri}
        assert canon(rank[t]) == canon(rank_ref[t_ref])
    n = 160
    x = torch.linspace(-1, 1, n, device=dev)
    gx, gy, gz = torch.meshgrid(x, x, x, indexing="ij")
    sphere = torch.sqrt(gx ** 2 + gy ** 2 + gz ** 2) - 0.7
    v, t = ana.mesh.marching_cubes(sphere.contiguous(), 0.0)
    vn, tn = v.cpu().numpy(), t.cpu().numpy()
    n_edges = check_closed_oriented_surface(vn, tn)
    assert vn.shape[0] - n_edges + tn.shape[0] == 2                      # a sphere
    p = vn / (n - 1) * 2 - 1
    assert np.abs(np.linalg.norm(p, axis=-1) - 0.7).max() < 1e-4
    vol = np.einsum("ij,ij->i", p[tn[:, 0]], np.cross(p[tn[:, 1]], p[tn[:, 2]])).sum() / 6
    assert abs(vol - 4 / 3 * np.pi * 0.7 ** 3) < 2e-3 * vol              # outward normals, the ball's volume
    empty_v, empty_t = ana.mesh.marching_cubes(torch.ones(5, 5, 5, device=dev), 0.0)
    assert empty_v.shape == (0, 3) and empty_t.shape == (0, 3)


# ----------------------------------------------------------------------------- configs[4] at its real size
def _grid_world(dev, smpl_table, mode):
    """BASELINE configs[4] as bench.py runs it: the seeded animated pose, sigma rescaled about its median, fine network."""
    from anim_nerf_amd import synthetic as syn
    m = seeded_model(smpl_table, 0, True, device=dev, mlp_mode=mode)
    with torch.no_grad():
        probe = (torch.rand(1, 4096, 3, generator=torch.Generator().manual_seed(5)) * 1.2 - 0.6)
        for net in (m.nerf, m.nerf_fine):                       # spread sigma about its median (literal init: one sign everywhere)
            s = orc.mlp_sigma_and_feature(net_params(net), probe)[0]
            net.sigma.weight.mul_(3000.0)
            net.sigma.bias.copy_((5.0 - 3000.0 * (s.median() - net.sigma.bias.cpu())).to(dev))
        pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=100).items()}
        rays = torch.zeros(1, 1, 8, device=dev)
        rays[..., 5], rays[..., 7] = -1, 10
        m.set_body_model(pose, _templ(dev))
        m.convert_to_body_model_space(rays)
        m.clac_ober2cano_transform()
    return m


def check_closed_oriented_surface_torch(faces, n_verts):
    """tests/test_mesh.py::check_closed_oriented_surface for millions of triangles: every directed edge once, and its opposite
    present (closed, consistently oriented 2-manifold).  Returns the number of undirected edges."""
    e = faces[:, [0, 1, 1, 2, 2, 0]].reshape(-1, 2)
    assert (e[:, 0] != e[:, 1]).all(), "degenerate triangle"
    key = torch.sort(e[:, 0] * n_verts + e[:, 1]).values
    assert (key[1:] != key[:-1]).all(), "an oriented edge used twice"
    assert torch.equal(key, torch.sort(e[:, 1] * n_verts + e[:, 0]).values), "an edge without its opposite: a hole or an orientation flip"
    return key.numel() // 2


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_sigma_grid_at_512_cubed(dev, smpl_table, mode):
    """extract_mesh.py:27-61,152-165 at BASELINE configs[4]'s real size, in the mode bench.py times (bf16) and in the parity
    mode: (a) the 2^27-point grid as ONE call (a 2.1 GB point tensor: byte offsets past 2^31) == the same grid in 2^22-point
    chunks, bit for bit; (b) on three 64^3 sub-blocks straddling the body the fast path (device grid, no search for
    provably-empty voxels, sigma-only MLP on the valid voxels) == the exact path (explicit points through AnimNeRF.forward,
    the reference's loop), bit for bit; f32: >= 20,000 sampled voxels against the oracle, voxel by voxel (tests/accounting.py);
    (c) marching cubes on the 512^3 volume: closed, consistently oriented, exactly one vertex per straddling grid edge."""
    import anim_nerf_amd as ana
    from accounting import account_for_points
    N, rng = 512, (-1.2, 1.2)
    m = _grid_world(dev, smpl_table, mode)
    one, first = ana.sigma_grid(m, N, rng, rng, rng, chunk=1 << 27)
    assert first == 0 and one.shape == (N ** 3,)
    parts = torch.cat([ana.sigma_grid(m, N, rng, rng, rng, chunk=1 << 22, rank=r, world=4)[0] for r in range(4)])
    assert torch.equal(one, parts), f"one call != chunked: {(one != parts).sum().item()} voxels differ"
    del parts
    occ = one > 0
    n_occ = int(occ.sum())
    assert 1e6 < n_occ < 2e7 and (one >= 0).all(), n_occ                  # a body in mostly empty space
    # (b) sub-blocks: the reference's create_grid (np.meshgrid 'xy': array axis 0 runs over y) + centre, explicit points
    lin = np.linspace(rng[0], rng[1], N)
    center = (m.verts.max(dim=1)[0] + m.verts.min(dim=1)[0]) / 2.
    vol = one.view(N, N, N)
    idx = torch.nonzero(occ.view(N, N, N))
    gen = torch.Generator().manual_seed(11)
    blocks = []
    for pick in torch.randint(0, idx.shape[0], (3,), generator=gen):
        a0, b0, c0 = [int(min(max(int(v) - 32, 0), N - 64)) for v in idx[pick]]
        blocks.append((a0, b0, c0))
        a, b, c = np.meshgrid(np.arange(a0, a0 + 64), np.arange(b0, b0 + 64), np.arange(c0, c0 + 64), indexing="ij")
        pts = np.stack([lin[b], lin[a], lin[c]], -1).reshape(-1, 3)                 # grid[a, b, c] = (x[b], y[a], z[c])
        points = torch.from_numpy(pts).unsqueeze(0).float().to(dev) + center
        exact = ana.sigma_grid_inference(m, points, chunk=32 * 32 * 64)[0, :, 0]
        fast = vol[a0:a0 + 64, b0:b0 + 64, c0:c0 + 64].reshape(-1)
        assert (exact > 0).any(), "the block must straddle the body"
        assert torch.equal(exact, fast), (blocks[-1], (exact != fast).sum().item(), (exact - fast).abs().max().item())
    if mode == "f32":
        # >= 20,000 voxels against the oracle: half of them occupied ones, half anywhere
        flat_occ = torch.nonzero(occ)[:, 0]
        sel = torch.cat([flat_occ[torch.randint(0, n_occ, (10240,), generator=gen).to(dev)],
                         torch.randint(0, N ** 3, (10240,), generator=gen).to(dev)])
        a, b, c = (sel // (N * N)).cpu().numpy(), ((sel // N) % N).cpu().numpy(), (sel % N).cpu().numpy()
        points = torch.from_numpy(np.stack([lin[b], lin[a], lin[c]], -1)).unsqueeze(0).float().to(dev) + center
        st = account_for_points(m, oracle_table(smpl_table), points, one[sel], use_fine=True, relu=True, label="sigma grid 512^3")
        assert st["valid"] >= 10000
    # (c) the level set extract_mesh.py takes next (:159-165)
    field = (5.0 - vol).contiguous()
    verts, tris = ana.mesh.marching_cubes(field, 0.0)
    inside = field < 0
    straddling = sum(int((inside.narrow(ax, 0, N - 1) != inside.narrow(ax, 1, N - 1)).sum()) for ax in range(3))
    assert verts.shape[0] == straddling, (verts.shape[0], straddling)
    edges = check_closed_oriented_surface_torch(tris, verts.shape[0])
    assert tris.shape[0] > 1e6 and 2 * edges == 3 * tris.shape[0]
    assert (verts >= 0).all() and (verts <= N - 1).all()
    print(f"\nsigma grid 512^3 [{mode}]: {n_occ} occupied voxels, blocks {blocks}, mesh {verts.shape[0]} vertices / {tris.shape[0]} triangles")


def test_timed_mode_mesh_within_one_voxel_of_the_parity_mode_mesh(dev, smpl_table):
    """configs[4] in the mode bench.py times (bf16) against the fp32 parity mode (itself held to the oracle voxel by voxel above)
    on the SAME 512^3 grid, as what extract_mesh.py:159-165 makes of it: the two level-set meshes, vertex by vertex, each
    against the other's SURFACE (point-to-triangle distances over the 125 cubes around a vertex, tests/accounting.py).  Gate:
    >= 99.7 % of the vertices of either mesh within ONE voxel of the other mesh and >= 99.9 % within two; the occupancy flips
    and the mean distance are printed and bounded.  (A sigma threshold turns bf16's ~1e-2 relative error into a surface that moves where sigma is
    flat: this is the number that says by how much.)"""
    import anim_nerf_amd as ana
    from accounting import vertex_to_surface_distance
    N, rng = 512, (-1.2, 1.2)
    meshes, occ = {}, {}
    for mode in ("f32", "bf16"):
        m = _grid_world(dev, smpl_table, mode)
        sig, _ = ana.sigma_grid(m, N, rng, rng, rng, chunk=1 << 27)
        occ[mode] = sig > 5.0                                           # extract_mesh.py:159: the level is sigma = 5
        meshes[mode] = ana.mesh.marching_cubes((5.0 - sig.view(N, N, N)).contiguous(), 0.0)
        del sig, m
    flips = int((occ["f32"] != occ["bf16"]).sum())
    n_occ = int(occ["f32"].sum())
    report = {"occupied_f32": n_occ, "occupancy_flips": flips}
    for a, b in (("bf16", "f32"), ("f32", "bf16")):
        d = vertex_to_surface_distance(meshes[a][0], meshes[b][0], meshes[b][1], N, reach=2)
        report[f"{a}_to_{b}"] = {"vertices": int(d.numel()), "within_1_voxel": round(float((d <= 1.0).float().mean()), 5),
                                 "within_2_voxels": round(float((d <= 2.0).float().mean()), 5),
                                 "within_half_voxel": round(float((d <= 0.5).float().mean()), 5), "mean_voxels": round(float(d.mean()), 4)}
    print(f"\nbf16 mesh vs f32 mesh, 512^3: {report}")
    # measured in round 5: 1.25 % of the occupied voxels flip, 99.81 / 99.84 % of the vertices within one voxel (the rest are
    # islands of a field that is flat around the level there: sigma = 3000 x a random-init network), mean distance 0.02 voxel
    assert flips <= 0.02 * n_occ, report
    for k in ("bf16_to_f32", "f32_to_bf16"):
        assert report[k]["within_1_voxel"] >= 0.997 and report[k]["within_2_voxels"] >= 0.999, report
        assert report[k]["mean_voxels"] <= 0.04, report
