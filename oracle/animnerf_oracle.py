"""CPU ORACLE for the Anim-NeRF per-ray rendering path.  TEST INFRASTRUCTURE ONLY.

This file restates, as plain functions over float32 CPU tensors, the algorithm
of the reference's hot path.  It is the checker for the HIP kernels: only
tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import
it.  The product (anim-nerf_amd/) never does, and fails loudly without its
HIP library.

Parity status: PINNED.  The reference holds no tests or golden vectors for
this path, so every function below is checked against outputs
of the reference itself, imported from /root/reference in the build container
by tests/golden/make_fixtures.py (and make_loss_fixtures.py for the training
side: NeRF.get_normal, AnimNeRFSystem.forward / compute_loss); the outputs are
committed under tests/golden/*.npz and compared in tests/test_oracle_golden.py.

Reference lines each function follows are cited in its docstring
(paths relative to the reference repository).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor
F32 = torch.float32

# ---------------------------------------------------------------------------
# a1  ray generation
# ---------------------------------------------------------------------------

def pixel_directions(H: int, W: int, focal, center=None) -> Tensor:
    """datasets/anim_nerf_dataset.py:56-70 — unit pinhole directions
    ((i-cx)/fx, -(j-cy)/fy, -1), no half-pixel offset.  [H,W,3]"""
    if center is None:
        center = (W * 0.5, H * 0.5)
    col = torch.linspace(0, W - 1, W, dtype=F32)[None, :].expand(H, W)
    row = torch.linspace(0, H - 1, H, dtype=F32)[:, None].expand(H, W)
    d = torch.stack([(col - center[0]) / focal[0],
                     -(row - center[1]) / focal[1],
                     -torch.ones(H, W, dtype=F32)], dim=-1)
    return d / torch.norm(d, dim=-1, keepdim=True)


def make_rays(c2w: Tensor, H: int, W: int, focal, near: float, far: float, center=None) -> Tensor:
    """datasets/anim_nerf_dataset.py:72-85 — rays[H,W,8] = [o(3), d(3), near, far]."""
    d_cam = pixel_directions(H, W, focal, center)
    d_world = d_cam @ c2w[:, :3].T
    o_world = c2w[:, 3].expand(d_world.shape)
    one = torch.ones_like(d_world[..., :1])
    return torch.cat([o_world, d_world, near * one, far * one], dim=-1)


def centred_pixel_directions(H: int, W: int, focal: float) -> Tensor:
    """utils/ray_utils.py:74-96 (get_ray_directions, the dead twin of pixel_directions): scalar focal,
    principal point at (W/2, H/2), normalised.  [H,W,3]"""
    col = torch.linspace(0, W - 1, W)[None, :].expand(H, W)
    row = torch.linspace(0, H - 1, H)[:, None].expand(H, W)
    d = torch.stack([(col - W / 2) / focal, -(row - H / 2) / focal, -torch.ones(H, W)], dim=-1)
    return d / torch.norm(d, dim=-1, keepdim=True)


def rotate_directions(directions: Tensor, c2w: Tensor) -> Tuple[Tensor, Tensor]:
    """utils/ray_utils.py:99-121 (get_rays) -> rays_o[H,W,3], rays_d[H,W,3]."""
    rays_d = directions @ c2w[:, :3].T
    return c2w[:, 3].expand(rays_d.shape), rays_d


# ---------------------------------------------------------------------------
# a2  SMPL forward / linear blend skinning
# ---------------------------------------------------------------------------

def axis_angle_to_matrix(rv: Tensor) -> Tensor:
    """smplx/lbs.py:298-332 — Rodrigues; note the angle is norm(rv + 1e-8)."""
    n = rv.shape[0]
    theta = torch.norm(rv + 1e-8, dim=1, keepdim=True)
    axis = rv / theta
    c = torch.cos(theta)[:, None]
    s = torch.sin(theta)[:, None]
    ax, ay, az = axis[:, 0:1], axis[:, 1:2], axis[:, 2:3]
    z = torch.zeros((n, 1), dtype=rv.dtype)
    K = torch.cat([z, -az, ay, az, z, -ax, -ay, ax, z], dim=1).view(n, 3, 3)
    eye = torch.eye(3, dtype=rv.dtype)[None]
    return eye + s * K + (1 - c) * torch.bmm(K, K)


def _rigid(R: Tensor, t: Tensor) -> Tensor:
    """smplx/lbs.py:335-345 — [R|t; 0 0 0 1]."""
    n = R.shape[0]
    top = torch.cat([R, t], dim=2)
    bottom = torch.tensor([0, 0, 0, 1], dtype=R.dtype).expand(n, 1, 4)
    return torch.cat([top, bottom], dim=1)


def kinematic_chain(rot: Tensor, joints: Tensor, parents: Tensor) -> Tuple[Tensor, Tensor]:
    """smplx/lbs.py:348-404 — world transform of each joint and the transform
    relative to the rest pose (A).  rot[B,J,3,3], joints[B,J,3]."""
    B, J = joints.shape[:2]
    jc = joints[..., None]
    rel = jc.clone()
    rel[:, 1:] = rel[:, 1:] - jc[:, parents[1:]]
    local = _rigid(rot.reshape(-1, 3, 3), rel.reshape(-1, 3, 1)).reshape(B, J, 4, 4)
    world = [local[:, 0]]
    for j in range(1, J):
        world.append(torch.matmul(world[int(parents[j])], local[:, j]))
    world = torch.stack(world, dim=1)
    posed = world[:, :, :3, 3]
    j_h = torch.cat([jc, torch.zeros_like(jc[:, :, :1])], dim=2)          # [B,J,4,1]
    shift = torch.matmul(world, j_h)                                     # [B,J,4,1]
    pad = torch.cat([torch.zeros(B, J, 4, 3, dtype=rot.dtype), shift], dim=3)
    return posed, world - pad


def smpl_forward(tbl: Dict[str, Tensor], betas: Tensor, global_orient: Tensor,
                 body_pose: Tensor, transl: Optional[Tensor]) -> Dict[str, Tensor]:
    """smplx/body_models.py:289-387 + smplx/lbs.py:152-251.

    tbl: v_template[V,3], shapedirs[V,3,10], posedirs[207,3V], J_regressor[24,V],
    parents[24], lbs_weights[V,24], extra_joints_idxs[21].
    Returns vertices, joints(45), joints_transform A, vertices_transform T,
    shape_offsets, pose_offsets — with transl added to vertices, joints AND the
    translation columns of A and T (body_models.py:370-374)."""
    B = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
    if betas.shape[0] != B:
        betas = betas.expand(B, -1)
    pose = torch.cat([global_orient, body_pose], dim=1)
    shape_off = torch.einsum('bl,mkl->bmk', betas, tbl['shapedirs'])
    v_shaped = tbl['v_template'] + shape_off
    J = torch.einsum('bik,ji->bjk', v_shaped, tbl['J_regressor'])
    R = axis_angle_to_matrix(pose.view(-1, 3)).view(B, -1, 3, 3)
    feat = (R[:, 1:] - torch.eye(3, dtype=R.dtype)).view(B, -1)
    pose_off = torch.matmul(feat, tbl['posedirs']).view(B, -1, 3)
    v_posed = pose_off + v_shaped
    Jp, A = kinematic_chain(R, J, tbl['parents'])
    nj = tbl['J_regressor'].shape[0]
    Wt = tbl['lbs_weights'][None].expand(B, -1, -1)
    T = torch.matmul(Wt, A.view(B, nj, 16)).view(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones_like(v_posed[..., :1])], dim=2)
    verts = torch.matmul(T, vh[..., None])[:, :, :3, 0]
    joints = torch.cat([Jp, verts[:, tbl['extra_joints_idxs']]], dim=1)
    if transl is not None:
        joints = joints + transl[:, None]
        verts = verts + transl[:, None]
        A = A.clone()
        T = T.clone()
        A[..., :3, 3] += transl[:, None]
        T[..., :3, 3] += transl[:, None]
    return dict(vertices=verts, joints=joints, joints_transform=A, vertices_transform=T,
                shape_offsets=shape_off, pose_offsets=pose_off)


# ---------------------------------------------------------------------------
# a3-a5  per-frame state
# ---------------------------------------------------------------------------

def apply_affine(M: Tensor, v: Tensor, w: float) -> Tensor:
    """models/anim_nerf.py:31-39 — (M @ [v, w])[:3] with broadcasting M[...,4,4]."""
    h = torch.full_like(v[..., :1], w)
    vh = torch.cat([v, h], dim=-1)
    return torch.matmul(M, vh[..., None])[..., :3, 0]


def frame_state(tbl, pose_params, template_params) -> Dict[str, Tensor]:
    """models/anim_nerf.py:108-126 (set_body_model)."""
    o = smpl_forward(tbl, **pose_params)
    t = smpl_forward(tbl, **template_params)
    nj = tbl['lbs_weights'].shape[1]
    return dict(
        verts=o['vertices'], joints=o['joints'][:, :nj], verts_transform=o['vertices_transform'],
        joints_transform=o['joints_transform'], shape_offsets=o['shape_offsets'],
        pose_offsets=o['pose_offsets'], global_transform=o['joints_transform'][:, 0].clone(),
        verts_template=t['vertices'], verts_transform_template=t['vertices_transform'],
        shape_offsets_template=t['shape_offsets'], pose_offsets_template=t['pose_offsets'])


def to_root_frame(st: Dict[str, Tensor], rays: Tensor) -> Tuple[Dict[str, Tensor], Tensor]:
    """models/anim_nerf.py:128-145 (convert_to_body_model_space).
    rays[bs,R,8] -> rays in the root-joint frame with near/far clamped to
    |o'| -/+ 1; verts, joints, global_transform, verts_transform move too."""
    Ginv = torch.inverse(st['global_transform'])[:, None]            # [bs,1,4,4]
    o = apply_affine(Ginv, rays[..., 0:3], 1.0)
    d = apply_affine(Ginv, rays[..., 3:6], 0.0)
    dist = torch.norm(o, dim=-1, keepdim=True)
    near = torch.max(rays[..., 6:7], dist - 1.0)
    far = torch.min(rays[..., 7:8], dist + 1.0)
    out = dict(st)
    out['verts'] = apply_affine(Ginv, st['verts'], 1.0)
    out['joints'] = apply_affine(Ginv, st['joints'], 1.0)
    out['global_transform'] = torch.matmul(Ginv[:, 0], st['global_transform'])
    out['verts_transform'] = torch.matmul(Ginv, st['verts_transform'])
    return out, torch.cat([o, d, near, far], dim=-1)


def observation_to_canonical(st: Dict[str, Tensor]) -> Tensor:
    """models/anim_nerf.py:147-151 (clac_ober2cano_transform):
    x_cano = T_template (T_pose^-1 x + (shape_t - shape) + (pose_t - pose))."""
    M = torch.inverse(st['verts_transform']).clone()
    M[..., :3, 3] += st['shape_offsets_template'] - st['shape_offsets']
    M[..., :3, 3] += st['pose_offsets_template'] - st['pose_offsets']
    return torch.matmul(st['verts_transform_template'], M)


# ---------------------------------------------------------------------------
# a8-a10  KNN + blend + warp
# ---------------------------------------------------------------------------

def knn_bruteforce(verts: Tensor, xyz: Tensor, k: int, chunk: int = 4096) -> Tuple[Tensor, Tensor]:
    """models/anim_nerf.py:161-163 — the in-repo definition of the external
    KNN_CUDA call (:159): Euclidean distance to every vertex, k smallest,
    ascending.  verts[bs,V,3], xyz[bs,N,3] -> dist[bs,N,k], idx[bs,N,k]."""
    ds, ids = [], []
    for s in range(0, xyz.shape[1], chunk):
        diff = xyz[:, s:s + chunk, None] - verts[:, None]
        d = torch.norm(diff, dim=-1, p=2)
        dk, ik = d.topk(k, largest=False, dim=-1)
        ds.append(dk)
        ids.append(ik)
    return torch.cat(ds, 1), torch.cat(ids, 1)


WEIGHT_STD = 0.1           # models/anim_nerf.py:84


def blend_neighbours(dist: Tensor, idx: Tensor, lbs_weights: Tensor, per_vertex_T: Tensor
                     ) -> Tuple[Tensor, Tensor, Tensor]:
    """models/anim_nerf.py:165-176.  Returns (blended distance [bs,N,1],
    blended 4x4 [bs,N,4,4], neighbour weights [bs,N,k])."""
    bs, V = per_vertex_T.shape[:2]
    std2 = 2.0 * WEIGHT_STD ** 2
    w_n = lbs_weights[idx]                                            # [bs,N,k,24]
    conf = torch.exp(-torch.sum(torch.abs(w_n - w_n[..., 0:1, :]), dim=-1) / std2)
    conf = (conf > 0.9).float()
    w = torch.exp(-dist) * conf
    w = w / w.sum(-1, keepdim=True)
    flat = per_vertex_T.reshape(bs * V, 4, 4)
    gidx = idx + (torch.arange(bs) * V)[:, None, None]
    Tn = flat[gidx]                                                   # [bs,N,k,4,4]
    Tb = torch.sum(w[..., None, None] * Tn, dim=2)
    db = torch.sum(w * dist, dim=2, keepdim=True)
    return db, Tb, w


def warp_to_canonical(xyz: Tensor, verts: Tensor, lbs_weights: Tensor, ober2cano: Tensor,
                      dis_threshold: float, k: int = 4, chunk: int = 4096):
    """models/anim_nerf.py:180-192 (unpose).  Returns xyz_c[bs,N,3],
    valid[bs,N,1] (float 0/1), and (dist, idx, blended_dist) for inspection.
    The neighbour search carries no gradient, as on the KNN_CUDA branch every config selects (:157-159)."""
    with torch.no_grad():
        dist, idx = knn_bruteforce(verts, xyz, k, chunk)
    db, Tb, _ = blend_neighbours(dist, idx, lbs_weights, ober2cano)
    valid = (db < dis_threshold).float()
    return apply_affine(Tb, xyz, 1.0), valid, dict(dist=dist, idx=idx, blended=db, transform=Tb)


def unpose_directions(viewdir: Tensor, blended_T: Tensor) -> Tensor:
    """models/anim_nerf.py:188-190 (unpose_view): batch_transform(xyz_transform_inv, viewdir) — with batch_transform's
    default pad_ones=True (:31-39), i.e. the direction is carried as a POINT (the translation is added)."""
    return apply_affine(blended_T, viewdir, 1.0)


# ---------------------------------------------------------------------------
# a11-a12  encoding + MLP
# ---------------------------------------------------------------------------

def fourier_encode(x: Tensor, n_freqs: int) -> Tensor:
    """models/embedding.py:22-39 — [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]."""
    parts = [x]
    for k in range(n_freqs):
        f = float(2 ** k)
        parts.append(torch.sin(f * x))
        parts.append(torch.cos(f * x))
    return torch.cat(parts, dim=-1)


def mlp_sigma_and_feature(P: Dict[str, Tensor], xyz: Tensor, n_freqs: int = 10):
    """models/nerf.py:155-175 — 8x256 ReLU trunk, skip at layer 5 with the
    encoding FIRST in the concat; sigma head and the 256-wide feature head have
    no activation.  P uses the reference's state-dict keys."""
    e = fourier_encode(xyz, n_freqs)
    h = e
    for i in range(8):
        if i == 4:
            h = torch.cat([e, h], dim=-1)
        h = torch.relu(torch.nn.functional.linear(
            h, P[f'xyz_encoding_{i+1}.0.weight'], P[f'xyz_encoding_{i+1}.0.bias']))
    sigma = torch.nn.functional.linear(h, P['sigma.weight'], P['sigma.bias'])
    feat = torch.nn.functional.linear(h, P['xyz_encoding_final.weight'], P['xyz_encoding_final.bias'])
    return sigma, feat


def mlp_forward(P: Dict[str, Tensor], xyz: Tensor, viewdir: Optional[Tensor] = None,
                n_freqs: int = 10, n_freqs_dir: int = 4, use_view: bool = False):
    """models/nerf.py:129-153 — rgb = sigmoid(W_r relu(W_d [feat, enc(dir)?]))."""
    sigma, feat = mlp_sigma_and_feature(P, xyz, n_freqs)
    x = feat
    if use_view:
        x = torch.cat([x, fourier_encode(viewdir, n_freqs_dir)], dim=-1)
    g = torch.relu(torch.nn.functional.linear(x, P['dir_encoding.0.weight'], P['dir_encoding.0.bias']))
    rgb = torch.sigmoid(torch.nn.functional.linear(g, P['rgb.0.weight'], P['rgb.0.bias']))
    return rgb, sigma


def mlp_forward_embedded(P: Dict[str, Tensor], input_xyz: Tensor, input_dir: Optional[Tensor] = None,
                         only_sigma: bool = False):
    """models/mlp.py:268-297 — the same network on an already embedded input (63 channels; the direction channels, if
    any, are concatenated behind the 256-wide feature)."""
    h = input_xyz
    for i in range(8):
        if i == 4:
            h = torch.cat([input_xyz, h], dim=-1)
        h = torch.relu(torch.nn.functional.linear(
            h, P[f'xyz_encoding_{i+1}.0.weight'], P[f'xyz_encoding_{i+1}.0.bias']))
    sigma = torch.nn.functional.linear(h, P['sigma.weight'], P['sigma.bias'])
    if only_sigma:
        return sigma
    x = torch.nn.functional.linear(h, P['xyz_encoding_final.weight'], P['xyz_encoding_final.bias'])
    if input_dir is not None and input_dir.shape[-1] > 0:
        x = torch.cat([x, input_dir], dim=-1)
    g = torch.relu(torch.nn.functional.linear(x, P['dir_encoding.0.weight'], P['dir_encoding.0.bias']))
    return torch.sigmoid(torch.nn.functional.linear(g, P['rgb.0.weight'], P['rgb.0.bias'])), sigma


def field_query(P, xyz, st, lbs_weights, use_unpose: bool, dis_threshold: float,
                chunk: int = 4096):
    """models/anim_nerf.py:290-307 (AnimNeRF.forward): warp, MLP, sigma=-1e5 where invalid."""
    if use_unpose:
        xyz_c, valid, _ = warp_to_canonical(xyz, st['verts'], lbs_weights, st['ober2cano'],
                                            dis_threshold, chunk=chunk)
    else:
        xyz_c, valid = xyz, torch.ones_like(xyz[..., :1])
    rgb, sigma = mlp_forward(P, xyz_c)
    sigma = torch.where(valid < 1, torch.full_like(sigma, -1e5), sigma)
    return rgb, sigma


# ---------------------------------------------------------------------------
# a6, a7, a13, a14  sampling + compositing
# ---------------------------------------------------------------------------

def coarse_depths(rays: Tensor, n_coarse: int, t_rand: Optional[Tensor] = None) -> Tensor:
    """models/volume_rendering.py:29-56 with lindisp=True:
    z_k = near (1 - s_k) + far s_k, s = linspace(0, 1 - 1/Kc, Kc); with `t_rand` (= perturb * U[0,1), shape of z)
    the stratified jitter of :48-54 — every depth moves inside [lower mid-point, upper mid-point]."""
    near, far = rays[..., 6:7], rays[..., 7:8]
    s = torch.linspace(0, 1 - 1.0 / n_coarse, n_coarse, dtype=rays.dtype)
    z = near * (1 - s) + far * s
    if t_rand is not None:
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


def composite(rgb: Tensor, sigma: Tensor, z: Tensor, far: Tensor, white_bkgd: bool = True):
    """models/volume_rendering.py:131-160.  rgb[...,K,3], sigma[...,K], z[...,K],
    far[...,1] -> weights[...,K], rgb[...,3], depth[...,1], acc[...,1]."""
    delta = torch.cat([z[..., 1:] - z[..., :-1], torch.full_like(z[..., :1], 1e10)], dim=-1)
    alpha = 1 - torch.exp(-delta * torch.relu(sigma))
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1 - alpha + 1e-10], -1), -1)
    w = alpha * trans[..., :-1]
    acc = w.sum(-1, keepdim=True)
    col = (w[..., None] * rgb).sum(-2)
    dep = (w * z).sum(-1, keepdim=True)
    if white_bkgd:
        dep = dep + (1 - acc) * far
        col = col + 1 - acc
    return w, col, dep, acc


def fine_depths(z_coarse: Tensor, weights: Tensor, n_fine: int, u: Optional[Tensor] = None,
                eps: float = 1e-5, details: bool = False):
    """models/volume_rendering.py:59-97,199-200 — inverse-CDF sampling over the
    Kc-1 mid-points with weights[1:-1]; u = linspace(0,1,Kf) when deterministic.
    details (a checker's option): also return, per sample, the cdf step `denom` before the `denom < eps -> 1` branch
    (:92-93), the distance `gap` of u to the nearer cdf entry of its bin (searchsorted's decision) — the two discontinuities
    a checker must be able to name — and the bin's `width` (d z / d cdf = width / denom: the sample's conditioning)."""
    Kc = z_coarse.shape[-1]
    bins = 0.5 * (z_coarse[..., :-1] + z_coarse[..., 1:])
    w = weights[..., 1:-1] + eps
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    if u is None:
        u = torch.linspace(0., 1., n_fine, dtype=z_coarse.dtype).expand(*z_coarse.shape[:-1], n_fine)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp_min(hi - 1, 0)
    hi = torch.clamp_max(hi, Kc - 2)
    c0, c1 = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b0, b1 = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    den_raw = c1 - c0
    den = torch.where(den_raw < eps, torch.ones_like(den_raw), den_raw)
    z = b0 + (u - c0) / den * (b1 - b0)
    if details:
        return z, dict(denom=den_raw, gap=torch.minimum((u - c0).abs(), (c1 - u).abs()), width=b1 - b0)
    return z


def render_rays(field, rays: Tensor, n_coarse: int, n_fine: int, white_bkgd: bool = True, z_fine: Optional[Tensor] = None,
                t_rand: Optional[Tensor] = None, noise=None, u: Optional[Tensor] = None):
    """models/volume_rendering.py:163-232, share_fine=False.
    `field(xyz[bs,N,3], use_fine) -> rgb[bs,N,3], sigma[bs,N,1]`.
    perturb = 0 by default; the training branches (perturb > 0) with their random numbers HANDED IN, so that a checker can
    feed both sides the same draws: t_rand[bs,R,Kc] = perturb * U[0,1) (the stratified jitter, :48-54), noise =
    (coarse[bs,R,Kc], fine[bs,R,Kc+Kf]) = noise_std * N(0,1) added to sigma before compositing (:128-129), u[bs,R,Kf] the
    importance sampler's uniforms (:66-70).
    z_fine (a checker's option, not the reference's): use these importance samples [bs,R,n_fine] instead of drawing them —
    the sampler is discontinuous (`denom < eps`, :92-93) and carries no gradient (:200), so a gradient check injects the
    samples of the path under test and compares everything that IS differentiated."""
    bs, R = rays.shape[:2]

    def shade(z, use_fine, nz):
        K = z.shape[-1]
        xyz = rays[..., None, :3] + z[..., None] * rays[..., None, 3:6]
        rgb, sig = field(xyz.reshape(bs, -1, 3), use_fine)
        sig = sig.reshape(bs, R, K)
        if nz is not None:
            sig = sig + nz.to(sig)
        return composite(rgb.reshape(bs, R, K, 3), sig, z, rays[..., 7:8], white_bkgd)

    zc = coarse_depths(rays, n_coarse, None if t_rand is None else t_rand.to(rays))
    w, col, dep, acc = shade(zc, False, None if noise is None else noise[0])
    out = dict(rgbs=col, alphas=acc, depths=dep, _z_coarse=zc, _weights=w)
    if n_fine > 0:
        # volume_rendering.py:200: no gradient through sampling
        zf = (fine_depths(zc, w.detach(), n_fine, None if u is None else u.to(zc)) if z_fine is None else z_fine.to(zc)).detach()
        zs, _ = torch.sort(torch.cat([zc, zf], -1), dim=-1)
        w2, col2, dep2, acc2 = shade(zs, True, None if noise is None else noise[1])
        out.update(rgbs_fine=col2, alphas_fine=acc2, depths_fine=dep2,
                   _z_fine=zf, _z_sorted=zs, _weights_fine=w2)
    return out


# ---------------------------------------------------------------------------
# a15  frame driver
# ---------------------------------------------------------------------------

def render_frame(tbl, P_coarse, P_fine, rays, pose_params, template_params, *, n_coarse, n_fine,
                 use_unpose, dis_threshold=0.2, chunk=512, white_bkgd=True, knn_chunk=2048, z_fine=None, t_rand=None,
                 noise=None, u=None):
    """train.py:189-215 / novel_view.py:78-98 — per-frame setup, ray-chunk loop, cat.
    (z_fine, t_rand, noise, u: see render_rays; per-ray tensors for the whole frame, chunked here.)"""
    st = frame_state(tbl, pose_params, template_params)
    st, rays_b = to_root_frame(st, rays)
    st['ober2cano'] = observation_to_canonical(st)

    def field(xyz, use_fine):
        return field_query(P_fine if use_fine else P_coarse, xyz, st, tbl['lbs_weights'],
                           use_unpose, dis_threshold, chunk=knn_chunk)

    pieces = []
    for s in range(0, rays_b.shape[1], chunk):
        cut = lambda t: None if t is None else t[:, s:s + chunk]
        pieces.append(render_rays(field, rays_b[:, s:s + chunk], n_coarse, n_fine, white_bkgd, cut(z_fine), cut(t_rand),
                                  None if noise is None else (cut(noise[0]), cut(noise[1])), cut(u)))
    out = {k: torch.cat([p[k] for p in pieces], 1) for k in pieces[0]}
    out['_rays_body'] = rays_b
    return out


# ---------------------------------------------------------------------------
# a16  losses + normals regulariser (training side; pinned by tests/golden/{normals,train_loss}.npz,
#      outputs of the reference's own NeRF.get_normal / AnimNeRFSystem.compute_loss: make_loss_fixtures.py)
# ---------------------------------------------------------------------------

def point_normals(P: Dict[str, Tensor], xyz: Tensor, delta: float = 0.02, create_graph: bool = True) -> Tensor:
    """models/nerf.py:177-190 (NeRF.get_normal) — d alpha / d xyz with alpha = 1 - exp(-delta relu(sigma)); the graph is
    kept so that a loss on the normals differentiates a second time w.r.t. the weights."""
    with torch.enable_grad():
        x = xyz if xyz.requires_grad else xyz.detach().clone().requires_grad_(True)
        sigma = mlp_sigma_and_feature(P, x)[0]
        alpha = 1 - torch.exp(-delta * torch.relu(sigma))
        return torch.autograd.grad(alpha, x, torch.ones_like(alpha), create_graph=create_graph, retain_graph=True)[0]


def training_loss(P_coarse, P_fine, results: Dict[str, Tensor], rgbs: Tensor, alphas: Tensor, *, n_samples: int,
                  fg_points: Optional[Tensor] = None, bg_points: Optional[Tensor] = None, verts_template: Optional[Tensor] = None,
                  draws: Optional[Tuple[Tensor, Tensor]] = None, use_unpose: bool = True, lambda_alphas: float = 0.1,
                  lambda_foreground: float = 0.01, lambda_background: float = 0.01, lambda_normals: float = 0.01,
                  epsilon: float = 0.01, dis_threshold: float = 0.2):
    """train.py:228-322 (AnimNeRFSystem.compute_loss) — rgb MSE, alpha L1, foreground / background sigma priors
    (query_canonical_space = the network in canonical space, models/anim_nerf.py:211-243) and the normals regulariser,
    coarse and — `P_fine` given, i.e. n_importance > 0 and not share_fine — fine.  `draws` = the two standard-normal tensors
    the reference takes from randn_like (train.py:289-290), handed in; None skips the term.  -> (total, details)."""
    mse, l1 = torch.nn.functional.mse_loss, torch.nn.functional.l1_loss
    fine = P_fine is not None
    d: Dict[str, Tensor] = {}
    d['loss_rgb'] = mse(results['rgbs'], rgbs)
    loss = d['loss_rgb']
    if fine:
        d['loss_rgb_fine'] = mse(results['rgbs_fine'], rgbs)
        loss = loss + d['loss_rgb_fine']
    d['loss_alphas'] = l1(results['alphas'], alphas)
    loss = loss + lambda_alphas * d['loss_alphas']
    if fine:
        d['loss_alphas_fine'] = l1(results['alphas_fine'], alphas)
        loss = loss + lambda_alphas * d['loss_alphas_fine']
    nets = (('', P_coarse),) + ((('_fine', P_fine),) if fine else ())
    k = -2.0 / n_samples
    if use_unpose and fg_points is not None:
        for tag, P in nets:
            d['loss_foreground' + tag] = torch.mean(torch.exp(k * torch.relu(mlp_sigma_and_feature(P, fg_points)[0])))
            loss = loss + lambda_foreground * d['loss_foreground' + tag]
    if use_unpose and bg_points is not None:
        for tag, P in nets:
            d['loss_background' + tag] = torch.mean(1 - torch.exp(k * torch.relu(mlp_sigma_and_feature(P, bg_points)[0])))
            loss = loss + lambda_background * d['loss_background' + tag]
    if draws is not None:
        points = verts_template.detach() + draws[0] * dis_threshold * 0.5
        neighbs = points + draws[1] * epsilon
        for tag, P in nets:
            n0, n1 = point_normals(P, points), point_normals(P, neighbs)
            n0 = n0 / (torch.norm(n0, p=2, dim=-1, keepdim=True) + 1e-5)
            n1 = n1 / (torch.norm(n1, p=2, dim=-1, keepdim=True) + 1e-5)
            d['loss_normals' + tag] = mse(n0, n1)
            loss = loss + lambda_normals * d['loss_normals' + tag]
    return loss, d


def psnr(a: Tensor, b: Tensor) -> float:
    """models/evaluator.py:18 — 10 log10(1 / MSE), data range 1."""
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float('inf') if mse == 0 else 10.0 * math.log10(1.0 / mse)
