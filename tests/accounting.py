"""Ray-by-ray accounting of a rendered frame against a reference render — the machinery behind the 1e-4 statement.

North-star tolerance: 1e-4 relative on every rendered value (fp32 mode).  The reference's path has three discontinuities:
the importance sampler's `denom < eps -> 1` branch (models/volume_rendering.py:92-93: a fine sample moves by up to a bin when
the cdf changes in its last ulp), the warp's validity threshold (models/anim_nerf.py:183) and the neighbour set itself (a tie,
or a blend-weight confidence within rounding of 0.9, models/anim_nerf.py:165-168).  A ray whose sorted depths, validity bits
or neighbour set differ from the reference's cannot be expected to meet 1e-4; every OTHER ray must.  So instead of a percentage
gate, `account_for_rays` takes every ray outside 1e-4 and re-renders it with the ORACLE fed with the HIP path's own decisions:

  1. sorted depths + validity bits injected            -> must agree within 1e-4, or
  2. canonical points injected as well                 -> must agree within 1e-4 (100 %), and every injected point that is not
                                                          the oracle's up to rounding must sit at a neighbour tie / confidence
                                                          threshold.

Anything left over is a real bug and fails the caller.  Used by tests/test_gpu_parity.py, __graft_entry__.smoke() and
bench.py's oracle_check; test infrastructure (imports the oracle)."""
import torch

from oracle import animnerf_oracle as orc

RTOL = 1e-4


def outside(a, b, rtol=RTOL, atol=1e-5):
    """[R] bool: rows of a[1,R,C] outside atol + rtol |b|."""
    return ((a - b).abs() > atol + rtol * b.abs()).any(-1)[0]


def render_stages(model, vr, rays_w, pose, templ, frame_setup=False):
    """One frame (bs = 1) through the HIP path stage by stage: the rendered tensors plus the sampling decisions behind
    them (coarse depths, sorted depths, validity bits).  Everything on the model's device."""
    import anim_nerf_amd as ana
    dev = next(model.parameters()).device
    warp = bool(model.use_unpose)
    Kc, Kf = vr.n_coarse, vr.n_fine
    with torch.no_grad():
        if frame_setup:                  # the set-up of the training steps (AnimNeRF.frame_setup: two launches, the same values to rounding)
            rays_b = model.frame_setup({k: v.to(dev) for k, v in pose.items()}, {k: v.to(dev) for k, v in templ.items()}, rays_w.to(dev))
        else:
            model.set_body_model({k: v.to(dev) for k, v in pose.items()}, {k: v.to(dev) for k, v in templ.items()})
            rays_b = model.convert_to_body_model_space(rays_w.to(dev))
            model.clac_ober2cano_transform()
        zc = vr.sample_coarse(rays_b)
        w_c, rgb_c, dep_c, acc_c = vr._shade(model, rays_b, zc, True, 0.0, True)
        out = dict(rgbs=rgb_c, alphas=acc_c, depths=dep_c)
        zs = valid_f = None
        if Kf:
            zs = vr.sample_fine_sorted(zc, w_c)
            _, rgb_f, dep_f, acc_f = vr._shade(model, rays_b, zs, False, 0.0, False)
            out.update(rgbs_fine=rgb_f, alphas_fine=acc_f, depths_fine=dep_f)
        valid_c = None
        if warp:
            valid_c = model.warped_points(rays=rays_b, z=zc)[:, 3].view(1, -1, Kc).cpu()
            if Kf:
                valid_f = model.warped_points(rays=rays_b, z=zs)[:, 3].view(1, -1, Kc + Kf).cpu()
    shape = lambda v: v.view(1, rays_w.shape[1], -1)
    return dict(out={k: shape(v) for k, v in out.items()}, rays_b=rays_b, zc=zc, zs=zs, w_c=w_c, valid_c=valid_c, valid_f=valid_f)


def importance_sample_excuse(z_hip, z_oracle, det, eps=1e-5, cdf_rounding=1e-6):
    """bool tensor: where a fine depth of the HIP path may differ from the reference sampler's on the same weights.  The cdf
    entries are fp32 sums next to 1 over a pdf normalised by a 62-term fp32 sum: they carry ~`cdf_rounding` of absolute
    rounding, whatever order the terms are added in.  That rounding (a) flips the `denom < eps -> 1` branch
    (models/volume_rendering.py:92-93) where denom is within it of eps, (b) flips searchsorted's bin where u is within it of a
    cdf entry, and (c) moves the interpolated depth by (rounding / denom) x bin width — nearly empty bins (denom just above
    eps = 1e-5) are ill-conditioned by construction.  Anything beyond 2e-5 + (c) that is not (a) or (b) is a bug."""
    den = det["denom"]
    allowed = 2e-5 + cdf_rounding / torch.where(den < eps, torch.ones_like(den), den) * det["width"].abs()
    off = (z_hip - z_oracle).abs() > allowed
    return off, ((den - eps).abs() <= cdf_rounding) | (det["gap"] <= cdf_rounding)


def check_importance_samples(vr, zc, w_c, zs, eps=1e-5):
    """The HIP path's importance samples ARE the reference sampler's output on the HIP path's own coarse weights, up to the
    conditioning `importance_sample_excuse` spells out.  Returns (samples, samples that moved at a named discontinuity)."""
    import anim_nerf_amd as ana
    bs, R, Kc = zc.shape
    Kf = vr.n_fine
    u = vr._table(zc.device, "u", Kf)
    zs2, zf = ana.ops.sample_fine_merge(zc.view(bs * R, Kc), w_c.view(bs * R, Kc), u, want_fine=True)
    assert torch.equal(zs2.view_as(zs), zs), "fused coarse pass and the stand-alone sampler disagree"
    zf_o, det = orc.fine_depths(zc.cpu(), w_c.view(bs, R, Kc).cpu(), Kf, details=True)
    off, excuse = importance_sample_excuse(zf.view(bs, R, Kf).cpu(), zf_o, det, eps)
    assert (off <= excuse).all(), (f"{int((off & ~excuse).sum())} importance samples differ from the reference sampler's on the "
                                   "same weights away from its discontinuities")
    return off.numel(), int(off.sum())


def account_for_rays(model, vr, smpl_tbl_oracle, rays_w, pose, templ, ref, *, stages=None, z_fine_ref=None, label="", dis_threshold=0.2,
                     quiet=False):
    """Hold one rendered frame (bs = 1, fp32 mode) to `ref` (the reference's — or the oracle's own — six rendered tensors):
    every ray within 1e-4, or accounted for as the module docstring says.  Returns the statistics.
    smpl_tbl_oracle = helpers.oracle_table(table); z_fine_ref (optional) = the reference's importance samples: then every
    accounted-for ray must also SHOW a cause (a depth that is not the reference's, a validity bit that is not the oracle's,
    or conditioning)."""
    import anim_nerf_amd as ana
    tbl = smpl_tbl_oracle
    st8 = stages or render_stages(model, vr, rays_w, pose, templ)
    warp = bool(model.use_unpose)
    Kc, Kf = vr.n_coarse, vr.n_fine
    got = {k: v.cpu() for k, v in st8["out"].items()}
    ref = {k: torch.as_tensor(ref[k]).view(got[k].shape) for k in got}
    rays_b, zc, zs, valid_c, valid_f = st8["rays_b"], st8["zc"], st8["zs"], st8["valid_c"], st8["valid_f"]
    R = rays_w.shape[1]
    assert torch.equal(zc.cpu(), orc.coarse_depths(rays_b.cpu(), Kc)), "coarse depths are deterministic: bit-exact or broken"
    keys_c = ("rgbs", "alphas", "depths")
    bad_c = torch.zeros(R, dtype=torch.bool)
    bad_f = torch.zeros(R, dtype=torch.bool)
    for k in keys_c:
        bad_c |= outside(got[k], ref[k])
        if Kf:
            bad_f |= outside(got[k + "_fine"], ref[k + "_fine"])
    bad = torch.nonzero(bad_c | bad_f)[:, 0]
    stats = dict(rays=R, outside_coarse=int(bad_c.sum()), outside_fine=int(bad_f.sum()), outside=int(bad.numel()),
                 max_abs_err=max((got[k] - ref[k]).abs().max().item() for k in got), after_depths_and_validity=0,
                 after_canonical_points=0, moved_points=0, moved_points_at_a_tie_or_threshold=0)
    if not quiet:
        print(f"\n{label}: {stats['outside_coarse']} coarse / {stats['outside_fine']} fine of {R} rays outside 1e-4 of the reference")
    if not warp:
        assert not bad_c.any(), "without the warp the coarse pass has no discontinuity: every ray must meet 1e-4"
    if Kf:                                # the sampler itself, on every ray (not only the out-of-tolerance ones)
        stats["importance_samples"], stats["importance_samples_moved_at_the_branch"] = check_importance_samples(vr, zc, st8["w_c"], zs)
    n = bad.numel()
    if n == 0:
        return stats

    # ---- the oracle on the out-of-tolerance rays, with the HIP path's decisions injected
    st = orc.frame_state(tbl, pose, templ)
    st, rays_o = orc.to_root_frame(st, rays_w)
    torch.testing.assert_close(rays_o, rays_b.cpu(), rtol=1e-5, atol=5e-6)
    st["ober2cano"] = orc.observation_to_canonical(st)
    from helpers import net_params
    Pc = net_params(model.nerf)
    Pf = net_params(model.nerf_fine) if Kf else None
    rb = rays_b.cpu()[:, bad]

    def oracle_pass(P, z, valid_hip, rows, xyz_c_hip=None):
        """the oracle's composite of rays `rows` (indices into `bad`) at depths z; warp on: validity bits from the HIP path,
        canonical points from the oracle's own warp or (xyz_c_hip) from the HIP path as well."""
        K, nr = z.shape[-1], rows.numel()
        rr = rb[:, rows]
        xyz = (rr[..., None, :3] + z[..., None] * rr[..., None, 3:6]).reshape(1, -1, 3)
        flips = torch.zeros(nr, dtype=torch.bool)
        if warp:
            xyz_c, valid_o, dbg = orc.warp_to_canonical(xyz, st["verts"], tbl["lbs_weights"], st["ober2cano"], dis_threshold, chunk=2048)
            flipped = (valid_o.view(1, nr, K) != valid_hip)
            flips = flipped.any(-1)[0]
            # a validity bit that is not the oracle's needs its cause: the blended neighbour distance within rounding of the
            # threshold (models/anim_nerf.py:183), or a neighbour tie / confidence threshold (:165-168) changing the blend
            w_n = tbl["lbs_weights"][dbg["idx"]]
            conf = torch.exp(-(w_n - w_n[..., 0:1, :]).abs().sum(-1) / (2.0 * orc.WEIGHT_STD ** 2))
            excuse = (((dbg["blended"][..., 0] - dis_threshold).abs() <= 1e-5) | ((conf - 0.9).abs() < 2e-6).any(-1)
                      | ((dbg["dist"][..., 1:] - dbg["dist"][..., :-1]).abs() <= 1e-6).any(-1)).view(1, nr, K)
            assert (flipped <= excuse).all(), f"{int((flipped & ~excuse).sum())} validity bits differ from the oracle's without a threshold to blame"
            if xyz_c_hip is not None:
                xyz_c = xyz_c_hip.reshape(1, -1, 3)
            rgb, sig = orc.mlp_forward(P, xyz_c)
            sig = torch.where(valid_hip.reshape(1, -1, 1) < 1, torch.full_like(sig, -1e5), sig)
        else:
            rgb, sig = orc.mlp_forward(P, xyz)
        _, col, dep, acc = orc.composite(rgb.view(1, nr, K, 3), sig.view(1, nr, K), z, rr[..., 7:8])
        return dict(rgbs=col, depths=dep, alphas=acc), flips

    def residual_of(rows, xyz_hip=(None, None)):
        oc, vflip = oracle_pass(Pc, zc.cpu()[:, bad[rows]], valid_c[:, bad[rows]] if warp else None, rows, xyz_hip[0])
        res = torch.zeros(rows.numel(), dtype=torch.bool)
        for k in keys_c:
            res |= outside(got[k][:, bad[rows]], oc[k])
        if Kf:
            of, vflip_f = oracle_pass(Pf, zs.cpu()[:, bad[rows]], valid_f[:, bad[rows]] if warp else None, rows, xyz_hip[1])
            vflip = vflip | vflip_f
            for k in keys_c:
                res |= outside(got[k + "_fine"][:, bad[rows]], of[k])
        return res, vflip
    residual, vflip = residual_of(torch.arange(n))
    stats["after_depths_and_validity"] = int(residual.sum())
    zflip = zdiff = torch.zeros(n, dtype=torch.bool)
    if Kf and z_fine_ref is not None:
        zs_ref = torch.sort(torch.cat([zc.cpu()[:, bad], torch.as_tensor(z_fine_ref)[:, bad]], -1), -1).values
        zflip = ((zs.cpu()[:, bad] - zs_ref).abs() > 2e-5).any(-1)[0]          # a sample that changed bins
        zdiff = (zs.cpu()[:, bad] != zs_ref).any(-1)[0]                        # any bit of any depth
    stats.update(moved_fine_depth=int(zflip.sum()), flipped_validity=int(vflip.sum()))
    if not quiet:
        print(f"{label}: of {n} rays: {int(zflip.sum())} with a moved fine depth ({int(zdiff.sum())} with depths that are not the "
              f"reference's bit for bit), {int(vflip.sum())} with a flipped validity bit, {int(residual.sum())} still outside 1e-4 "
              f"of the oracle given the HIP path's depths / validity")
    if not warp:
        assert not residual.any(), "rays that differ from the oracle even with identical sampling decisions: a real bug"
        if z_fine_ref is not None:
            # (a depth that is not the reference's: moved by a bin at the sampler's branch, or by its conditioning in a nearly
            # empty bin — at a sigma gain of thousands a 1e-5 move of a sample on a density edge is visible)
            assert zdiff.all(), "out-of-tolerance rays whose sorted depths ARE the reference's, bit for bit: nothing to blame"
        return stats
    # Warp on: what is left is the conditioning of the canonical coordinates (tests/test_oracle_golden.py::
    # test_reference_conditioning: a 1-ulp move of the sample points moves the reference itself by > 1e-4 on some rays; the
    # 2^9 Fourier band and the sigma gain amplify ~1e-7 of fp32 rounding in the 4x4 inverses and blends).  Third injection
    # for exactly those rays: the HIP path's canonical points.  Then (a) MLP + compositing must agree with the oracle within
    # 1e-4 on every one of them, and (b) the injected points must be the oracle's up to fp32 rounding — or sit at a tie.
    rows = torch.nonzero(residual)[:, 0]
    if rows.numel():
        sub = rays_b[:, bad[rows]].contiguous()
        wargs = (model.knn_index(), model.ober2cano_transform, model.body_model.lbs_weights, dis_threshold)
        passes = [(ana.ops.warp_points(*wargs, rays=sub, z=zc[:, bad[rows]].contiguous(), debug=True), zc)]
        if Kf:
            passes.append((ana.ops.warp_points(*wargs, rays=sub, z=zs[:, bad[rows]].contiguous(), debug=True), zs))
        res3, _ = residual_of(rows, tuple(p[0][0][..., :3].cpu() for p in passes) + ((None,) if not Kf else ()))
        moved = blamed = 0
        worst_plain = 0.0
        for (pts_h, dist_h, idx_h, _), z_ in passes:
            zz = z_.cpu()[:, bad[rows]]
            rr = rb[:, rows]
            xyz = (rr[..., None, :3] + zz[..., None] * rr[..., None, 3:6]).reshape(1, -1, 3)
            xyz_c, valid_o, dbg = orc.warp_to_canonical(xyz, st["verts"], tbl["lbs_weights"], st["ober2cano"], dis_threshold, chunk=2048)
            w_n = tbl["lbs_weights"][dbg["idx"]]
            conf = torch.exp(-(w_n - w_n[..., 0:1, :]).abs().sum(-1) / (2.0 * orc.WEIGHT_STD ** 2))
            near_threshold = ((conf - 0.9).abs() < 2e-6).any(-1)[0]
            other_order = (idx_h.cpu().long() != dbg["idx"]).any(-1)[0]
            assert ((dist_h.cpu() - dbg["dist"]).abs() <= 1e-6 + 1e-5 * dbg["dist"]).all(), "neighbour distances differ"
            dx = (pts_h[..., :3].cpu() - xyz_c).abs().max(-1).values[0]
            live = (valid_o[0, :, 0] >= 1) & (pts_h[0, :, 3].cpu() >= 1)
            big = live & (dx > 5e-6)
            moved += int(big.sum())
            blamed += int((big & (near_threshold | other_order)).sum())
            worst_plain = max(worst_plain, dx[live & ~big].max().item() if (live & ~big).any() else 0.0)
            assert (big <= (near_threshold | other_order)).all(), "canonical points moved without a tie or a threshold to blame"
        stats.update(after_canonical_points=int(res3.sum()), moved_points=moved, moved_points_at_a_tie_or_threshold=blamed,
                     canonical_point_rounding=worst_plain)
        if not quiet:
            print(f"{label}: the {rows.numel()} remaining rays with the HIP path's canonical points injected as well: "
                  f"{int(res3.sum())} outside 1e-4; {moved} samples moved by > 5e-6 ({blamed} at a neighbour tie / confidence "
                  f"threshold), the others within {worst_plain:.1e}")
        assert not res3.any(), "MLP / compositing differ from the oracle on identical canonical points: a real bug"
    if z_fine_ref is not None:
        assert (zdiff | vflip | residual).all(), "out-of-tolerance rays without a discontinuity or conditioning to blame"
    return stats


def account_for_points(model, smpl_tbl_oracle, xyz, sigma, rgb=None, *, use_fine=True, relu=False, dis_threshold=0.2, label="",
                       quiet=False):
    """Point queries (AnimNeRF.forward as extract_mesh.py:49-61 calls it; bs = 1, fp32 mode) against the oracle, point by
    point: sigma (or relu(sigma) — `relu`) and rgb within 1e-4, or
      * the validity bit differs and the blended neighbour distance is within rounding of dis_threshold, or the neighbour
        set sits at a tie / confidence threshold (the reference's own discontinuities, models/anim_nerf.py:165-184), or
      * with the HIP path's canonical point injected the oracle's MLP agrees within 1e-4, and that point is the oracle's up
        to fp32 rounding (5e-6) or sits at a tie / confidence threshold.
    sigma: 1e-4 |sigma| + 1e-4 median|sigma| (it crosses zero).  Returns statistics."""
    import anim_nerf_amd as ana
    from helpers import net_params
    tbl = smpl_tbl_oracle
    P = net_params(model.nerf_fine if use_fine else model.nerf)
    xyz_cpu = xyz.cpu()
    n = xyz_cpu.shape[1]
    st = dict(verts=model.verts.cpu(), ober2cano=model.ober2cano_transform.cpu())
    xyz_c, valid_o, dbg = orc.warp_to_canonical(xyz_cpu, st["verts"], tbl["lbs_weights"], st["ober2cano"], dis_threshold, chunk=1024)
    rgb_o, sig_o = orc.mlp_forward(P, xyz_c)
    sig_o = torch.where(valid_o < 1, torch.full_like(sig_o, -1e5), sig_o)[0, :, 0]
    sig_h = sigma.reshape(-1).cpu()
    w_n = tbl["lbs_weights"][dbg["idx"]]
    conf = torch.exp(-(w_n - w_n[..., 0:1, :]).abs().sum(-1) / (2.0 * orc.WEIGHT_STD ** 2))
    at_tie = (((conf - 0.9).abs() < 2e-6).any(-1) | ((dbg["dist"][..., 1:] - dbg["dist"][..., :-1]).abs() <= 1e-6).any(-1))[0]
    at_threshold = ((dbg["blended"][0, :, 0] - dis_threshold).abs() <= 1e-5)
    inval_o = valid_o[0, :, 0] < 1
    if relu:
        # the valid flag is not observable through relu(sigma) where sigma <= 0: a flip shows only as a value difference
        want = torch.relu(sig_o)
        scale = want[want > 0].abs().median() if (want > 0).any() else torch.tensor(1.0)
        differs = (sig_h - want).abs() > 1e-4 * want.abs() + 1e-4 * scale
        flipped = differs & (inval_o | (sig_h == 0))
    else:
        inval_h = sig_h == -1e5
        flipped = inval_h != inval_o
        live = ~(inval_h | inval_o)
        scale = sig_o[live].abs().median() if live.any() else torch.tensor(1.0)
        differs = live & ((sig_h - sig_o).abs() > 1e-4 * sig_o.abs() + 1e-4 * scale)
        if rgb is not None:
            rgb_h = rgb.reshape(-1, 3).cpu()
            differs |= live & ((rgb_h - rgb_o[0]).abs() > 1e-5 + RTOL * rgb_o[0].abs()).any(-1)
    stats = dict(points=n, valid=int((~inval_o).sum()), differ=int(differs.sum()), validity_flips=int(flipped.sum()))
    # second look at every differing point: the HIP path's canonical point through the oracle's MLP
    rows = torch.nonzero(differs)[:, 0]
    unexplained = 0
    if rows.numel():
        dev = xyz.device
        pts_h, dist_h, idx_h, blended_h = ana.ops.warp_points(model.knn_index(), model.ober2cano_transform, model.body_model.lbs_weights,
                                                             dis_threshold, xyz=xyz[:, rows.to(dev)].contiguous(), debug=True)
        pts_h = pts_h.cpu()[0]
        rgb_i, sig_i = orc.mlp_forward(P, pts_h[None, :, :3])
        sig_i = torch.where(pts_h[:, 3] < 1, torch.full_like(sig_i[0, :, 0], -1e5), sig_i[0, :, 0])
        got = sig_h[rows]
        want_i = torch.relu(sig_i) if relu else sig_i
        ok = (got - want_i).abs() <= 1e-4 * want_i.abs() + 1e-4 * scale
        if rgb is not None and not relu:
            ok &= ((rgb.reshape(-1, 3).cpu()[rows] - rgb_i[0]).abs() <= 1e-5 + RTOL * rgb_i[0].abs()).all(-1) | (pts_h[:, 3] < 1)
        assert ok.all(), f"{label}: {int((~ok).sum())} points differ from the oracle's MLP on the HIP path's own canonical points: a real bug"
        moved = (pts_h[:, :3] - xyz_c[0, rows]).abs().max(-1).values > 5e-6
        vflip = (pts_h[:, 3] < 1) != inval_o[rows]
        excused = (~moved & ~vflip) | at_tie[rows] | (vflip & at_threshold[rows])
        unexplained = int((~excused).sum())
        stats.update(moved_points=int(moved.sum()), conditioning_only=int((~moved & ~vflip).sum()))
        assert unexplained == 0, f"{label}: {unexplained} points differ without a threshold, a tie or rounding to blame"
    if not quiet:
        print(f"\n{label}: {stats}")
    return stats


def neighbour_discontinuity(lbs_weights, dist, idx):
    """[bs,N] bool — the point sits at one of the discontinuities of the reference's blend (models/anim_nerf.py:157-176): two
    neighbour distances tied within 1e-6 (the neighbour ORDER — hence neighbour 0, the confidence anchor — is arbitrary), or a
    blend-weight confidence exp(-|w_k - w_0|_1 / 0.02) within 2e-6 of its 0.9 threshold.  dist / idx[bs,N,k] as the
    reference's (or the oracle's) exact search returns them."""
    w_n = lbs_weights[idx]
    conf = torch.exp(-(w_n - w_n[..., 0:1, :]).abs().sum(-1) / (2.0 * orc.WEIGHT_STD ** 2))
    return ((conf - 0.9).abs() < 2e-6).any(-1) | ((dist[..., 1:] - dist[..., :-1]).abs() <= 1e-6).any(-1)


def point_triangle_distance(p, a, b, c):
    """[n] distances of points p[n,3] to triangles (a, b, c)[n,3] (Ericson, Real-Time Collision Detection 5.1.5: the closest
    point by Voronoi region of the triangle), branch-free over tensors."""
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    tiny = 1e-30
    denom = (va + vb + vc).clamp_min(tiny)
    v, w = vb / denom, vc / denom
    q = a + ab * v[:, None] + ac * w[:, None]                                        # interior
    t_bc = ((d4 - d3) / ((d4 - d3) + (d5 - d6)).clamp_min(tiny)).clamp(0, 1)
    q = torch.where(((va <= 0) & (d4 - d3 >= 0) & (d5 - d6 >= 0))[:, None], b + (c - b) * t_bc[:, None], q)
    t_ac = (d2 / (d2 - d6).clamp_min(tiny)).clamp(0, 1)
    q = torch.where(((vb <= 0) & (d2 >= 0) & (d6 <= 0))[:, None], a + ac * t_ac[:, None], q)
    t_ab = (d1 / (d1 - d3).clamp_min(tiny)).clamp(0, 1)
    q = torch.where(((vc <= 0) & (d1 >= 0) & (d3 <= 0))[:, None], a + ab * t_ab[:, None], q)
    q = torch.where(((d6 >= 0) & (d5 <= d6))[:, None], c, q)
    q = torch.where(((d3 >= 0) & (d4 <= d3))[:, None], b, q)
    q = torch.where(((d1 <= 0) & (d2 <= 0))[:, None], a, q)
    return (p - q).norm(dim=-1)


def vertex_to_surface_distance(verts, other_verts, other_tris, N, reach=1, chunk=1 << 18, cap=None):
    """[n] distance (in voxels, clamped to `cap` = reach + 1 where nothing is near) from each vertex of one marching-cubes
    mesh to the SURFACE of another one of the same N^3 grid (vertices in index space, extract_mesh.py:159-165).  A marching-
    cubes triangle lies inside one grid cube, so the triangles that can be within `reach` voxels of a vertex are those of the
    (2 reach + 1)^3 cubes around it: triangles are sorted by their cube, each cube holds at most 5 (csrc/mesh.hip: MAX_TRIS 8
    bounds the table)."""
    dev = verts.device
    cap = float(reach + 1) if cap is None else float(cap)
    tv = other_verts[other_tris.long()]                                              # [T, 3, 3]
    cube = tv.mean(1).floor().clamp_(0, N - 2).long()
    key = (cube[:, 0] * N + cube[:, 1]) * N + cube[:, 2]
    key, order = torch.sort(key)
    tv = tv[order]
    per_cube = int(torch.unique_consecutive(key, return_counts=True)[1].max()) if key.numel() else 0
    out = torch.full((verts.shape[0],), cap, dtype=torch.float32, device=dev)
    offs = torch.arange(-reach, reach + 1, device=dev)
    offs = torch.stack(torch.meshgrid(offs, offs, offs, indexing="ij"), -1).reshape(-1, 3)
    for s in range(0, verts.shape[0], chunk):
        p = verts[s:s + chunk]
        base = p.floor().long()
        best = torch.full((p.shape[0],), cap, dtype=torch.float32, device=dev)
        for o in offs:
            cell = base + o
            inside = ((cell >= 0) & (cell <= N - 2)).all(-1)
            k = (cell[:, 0] * N + cell[:, 1]) * N + cell[:, 2]
            lo = torch.searchsorted(key, k)
            hi = torch.searchsorted(key, k, right=True)
            for j in range(per_cube):
                has = inside & (lo + j < hi)
                if not bool(has.any()):
                    break
                t = tv[(lo + j).clamp_max(max(tv.shape[0] - 1, 0))]
                d = point_triangle_distance(p, t[:, 0], t[:, 1], t[:, 2])
                best = torch.where(has, torch.minimum(best, d), best)
        out[s:s + chunk] = best
    return out
