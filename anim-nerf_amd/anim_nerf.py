"""`AnimNeRF` — the animatable field, with the reference's constructor, attributes and methods
(models/anim_nerf.py:41-307) and its per-point work on the HIP library.

Per frame (host, small):   set_body_model -> convert_to_body_model_space -> clac_ober2cano_transform
Per point  (HIP kernels):  forward(xyz) = warp (exact 4-NN + blend) -> fused encode+MLP -> sigma mask
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .body_model import SMPL, create, small_matmul, small_matvec
from .nerf import NeRF


def batch_transform(P, v, pad_ones=True):
    """(P @ [v, 1|0])[:3]   (models/anim_nerf.py:31-39)."""
    if torch.is_grad_enabled() and (P.requires_grad or v.requires_grad):
        out = small_matvec(P[..., :3, :3], v)             # training: no batched-GEMM call per tiny product
    else:
        out = (P[..., :3, :3] @ v[..., None])[..., 0]
    return out + P[..., :3, 3] if pad_ones else out


def _affine_inverse(T):
    """Inverse of affine 4x4 matrices [..., 4, 4] with bottom row (0,0,0,1) in closed form (adjugate / determinant):
    differentiable tensor ops only — no LAPACK call, nothing that reads the device back (graph-capturable)."""
    R, t = T[..., :3, :3], T[..., :3, 3]
    c = torch.linalg.cross
    r0, r1, r2 = R[..., 0, :], R[..., 1, :], R[..., 2, :]
    adj = torch.stack([c(r1, r2), c(r2, r0), c(r0, r1)], dim=-1)
    det = (r0 * c(r1, r2)).sum(-1)
    Rinv = adj / det[..., None, None]
    out = torch.zeros_like(T)
    out[..., :3, :3] = Rinv
    out[..., :3, 3] = -small_matvec(Rinv, t)
    out[..., 3, 3] = 1
    return out


def _ober2cano_autograd(T, T_template, offset_delta):
    """models/anim_nerf.py:147-151 as differentiable tensor ops (closed-form inverse of the affine T, per frame):
    out = T_template @ [R^-1 | -R^-1 t + offset_delta]."""
    R, t = T[..., :3, :3], T[..., :3, 3]
    c = torch.linalg.cross
    r0, r1, r2 = R[..., 0, :], R[..., 1, :], R[..., 2, :]
    adj = torch.stack([c(r1, r2), c(r2, r0), c(r0, r1)], dim=-1)          # columns = cofactor rows -> adjugate
    det = (r0 * c(r1, r2)).sum(-1)
    Rinv = adj / det[..., None, None]
    tinv = -small_matvec(Rinv, t) + offset_delta
    M = torch.zeros_like(T)
    M[..., :3, :3] = Rinv
    M[..., :3, 3] = tinv
    M[..., 3, 3] = 1
    return small_matmul(T_template, M)


def on_same_device(a, b):
    return a is not None and b is not None and a.device == b.device


class AnimNeRF(nn.Module):
    def __init__(self, model_path="smplx/models", model_type="smpl", gender="male", freqs_xyz=10, freqs_dir=4,
                 use_view=False, use_unpose=False, unpose_view=False, k_neigh=4, use_knn=False,
                 use_deformation=False, deformation_dim=0, apperance_dim=0, use_fine=False, share_fine=False,
                 dis_threshold=0.2, query_inside=False, body_model_table=None, mlp_mode: Optional[str] = None,
                 **kwargs):
        super().__init__()
        self.freqs_xyz, self.freqs_dir = freqs_xyz, freqs_dir
        self.use_view, self.use_unpose, self.unpose_view = use_view, use_unpose, unpose_view
        self.k_neigh, self.use_knn = k_neigh, use_knn
        self.use_deformation, self.deformation_dim, self.apperance_dim = use_deformation, deformation_dim, apperance_dim
        self.use_fine, self.share_fine = use_fine, share_fine
        self.dis_threshold = dis_threshold
        self.query_inside = query_inside
        if use_deformation:
            raise NotImplementedError("use_deformation is False in every shipped config and broken in the reference "
                                      "(models/nerf.py:54)")
        if not 1 <= k_neigh <= 8:
            raise NotImplementedError("k_neigh = 1..8 (the shipped configs use 4)")

        # `body_model_table` (a dict / SyntheticSMPL in SMPL-pickle layout) replaces the licensed file
        if body_model_table is not None:
            self.body_model = SMPL(data_struct=body_model_table, gender=gender)
        else:
            self.body_model = create(model_path, model_type, gender=gender)
        self.weight_std = 0.1
        self.lbs_dim = self.body_model.lbs_weights.shape[1]
        # fixed spatial order of the vertices for the per-frame KNN index (not part of the state dict)
        self.register_buffer("knn_order", ops.morton_order(self.body_model.v_template), persistent=False)
        self._knn_index = None
        self._refine = None
        self.skip_far_samples = True     # renderer only: no neighbour search for provably-invalid samples
        # renderer only: the MLP runs on the samples within dis_threshold of the body (the rest is sigma = -1e5 and
        # composites with weight exactly 0).  `query_inside=True` asks for the same at the forward() level, where it
        # also zeroes rgb of the other samples (models/anim_nerf.py:245-290).
        self.skip_invalid_samples = True
        if k_neigh != 4:
            # no shipped config: served by an exhaustive exact search (anr_knn_k) + the blend as tensor ops
            # (`_warp_generic`), without the renderer's far-sample pruning
            self.skip_far_samples = False

        mk = dict(freqs_xyz=freqs_xyz, freqs_dir=freqs_dir, use_view=use_view, deformation_dim=deformation_dim,
                  apperance_dim=apperance_dim, mlp_mode=mlp_mode)
        self.nerf = NeRF(**mk)
        if use_fine:
            self.nerf_fine = self.nerf if share_fine else NeRF(**mk)

    @property
    def evaluate_valid_only(self) -> bool:
        return bool(self.use_unpose and (self.skip_invalid_samples or self.query_inside))

    # ------------------------------------------------------------------ per-frame state
    def set_latent_code(self, latent_code):
        if self.deformation_dim > 0:
            self.deformation_code = latent_code[:, :self.deformation_dim]
            if self.apperance_dim > 0:
                self.apperance_code = latent_code[:, self.deformation_dim:self.deformation_dim + self.apperance_dim]
        elif self.apperance_dim > 0:
            self.apperance_code = latent_code[:, :self.apperance_dim]

    def set_body_model(self, body_model_params, body_model_params_template=None):
        # Pose refinement (optim_body_params, train.py:141-144): the values still come from the forward kernels; the
        # gradient of the whole per-frame chain is attached to its two consumed outputs (rays in the body frame,
        # ober2cano) by autograd.FrameChainFunction / anr_frame_backward.
        self._refine = None
        on_gpu = self.body_model.v_template.is_cuda
        if on_gpu and torch.is_grad_enabled() and any(torch.is_tensor(v) and v.requires_grad for v in body_model_params.values()):
            self._refine = dict(body_model_params)
            body_model_params = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in body_model_params.items()}
        o = self.body_model(**body_model_params, return_verts=True)
        self.verts = o["vertices"]
        self.joints = o["joints"][:, :self.lbs_dim]
        self.verts_transform = o["vertices_transform"]
        self.joints_transform = o["joints_transform"]
        self.shape_offsets = o["shape_offsets"]
        self.pose_offsets = o["pose_offsets"]
        self.global_transform = o["joints_transform"][:, 0].clone()
        self._knn_index = None
        self._o2c_attached = None
        if body_model_params_template is not None and not self._same_template(body_model_params_template):
            self._set_template(body_model_params_template)

    def frame_setup(self, body_model_params, body_model_params_template, rays):
        """set_body_model + convert_to_body_model_space + clac_ober2cano_transform for a batch of frames, -> the rays in the
        root-joint frame.  On the GPU, in the kernels' forms, the three are TWO launches (ops.frame_setup, csrc/frame_setup.hip:
        eight before); anything else (CPU, gradients that must flow through torch's own autograd, another body model) takes the
        three calls.  Both training steps — the explicit one and the autograd one — set their frames up through the same
        kernels, so they render the same bits."""
        p = body_model_params
        names = ("betas", "global_orient", "body_pose", "transl")
        bm = self.body_model
        import os
        fast = (not os.environ.get("ANR_FRAME_SETUP_OFF") and rays.is_cuda and bm.v_template.is_cuda and body_model_params_template is not None
                and all(torch.is_tensor(p.get(k)) and p[k].is_cuda and p[k].dtype == torch.float32 for k in names)
                and bm.lbs_weights.shape[1] == 24 and bm.shapedirs.shape[-1] == 10 and p["betas"].shape[-1] == 10
                and p["body_pose"].shape[-1] == 69 and rays.shape[-1] == 8 and rays.dim() == 3
                and not any(torch.is_tensor(v) and v.requires_grad for v in body_model_params_template.values()))
        if not fast:
            self.set_body_model(body_model_params, body_model_params_template)
            rays = self.convert_to_body_model_space(rays)
            self.clac_ober2cano_transform()
            return rays
        self._refine = None
        if torch.is_grad_enabled() and any(p[k].requires_grad for k in names):
            self._refine = dict(p)                              # (pose refinement: the chain's backward is attached below)
        if not self._same_template(body_model_params_template):
            with torch.no_grad():
                self._set_template(body_model_params_template)
        bs = rays.shape[0]
        with torch.no_grad():
            arrays = (p["betas"].detach().expand(bs, -1).contiguous(), p["global_orient"].detach().expand(bs, -1).contiguous(),
                      p["body_pose"].detach().expand(bs, -1).contiguous(), p["transl"].detach().expand(bs, -1).contiguous())
            o = ops.frame_setup(arrays, None, self._chain_consts(), bm,
                                (self.verts_transform_template, self.shape_offsets_template, self.pose_offsets_template), rays.detach())
        self.shape_offsets, self.pose_offsets, self.joints_transform = o["shape_offsets"], o["pose_offsets"], o["A"]
        self.global_transform, self.verts, self.joints, self.verts_transform = o["g_root"], o["verts"], o["joints"], o["verts_transform"]
        self._knn_index = None
        self._o2c_attached = None
        new_rays, o2c = o["rays_body"], o["ober2cano"]
        if self._refine is not None:
            new_rays, o2c = self._attach_chain(new_rays, rays.detach(), o2c)
        self.ober2cano_transform = o2c
        return new_rays

    def _set_template(self, body_model_params_template):
        t = self.body_model(**body_model_params_template, return_verts=True)
        self.verts_template = t["vertices"]
        self.joints_template = t["joints"][:, :self.lbs_dim]
        self.verts_transform_template = t["vertices_transform"]
        self.joints_transform_template = t["joints_transform"]
        self.shape_offsets_template = t["shape_offsets"]
        self.pose_offsets_template = t["pose_offsets"]

    def _same_template(self, template):
        """The template pose is one dict for a whole run (datasets/anim_nerf_dataset.py hands the same tensors to every
        step): its body state is computed once, not once per step.  The cache holds the tensors themselves (identity +
        version counter), so a changed or replaced template is always recomputed."""
        ref = getattr(self, "_template_ref", None)
        tensors = [v for v in template.values() if torch.is_tensor(v)]
        cacheable = all(v.is_cuda and not v.requires_grad for v in tensors)
        same = (cacheable and ref is not None and on_same_device(getattr(self, "verts_template", None), self.body_model.v_template)
                and ref.keys() == template.keys()
                and all((v is ref[k][0] and v._version == ref[k][1]) if torch.is_tensor(v) else v == ref[k][0]
                        for k, v in template.items()))
        self._template_ref = {k: (v, v._version if torch.is_tensor(v) else None) for k, v in template.items()} if cacheable else None
        return same

    def _pose_grad(self):
        """True when gradients must reach the SMPL parameters through the tensor-op forms (CPU, or parameters that do not
        come as a set_body_model dict); on the GPU the kernels + FrameChainFunction serve pose refinement."""
        return torch.is_grad_enabled() and self.verts_transform.requires_grad

    def _chain_consts(self):
        bm = self.body_model
        dev = bm.v_template.device
        c = getattr(self, "_chain_const_cache", None)
        if c is None or c["J0"].device != dev:
            with torch.no_grad():
                c = dict(J0=(bm.J_regressor @ bm.v_template).contiguous(),
                         JS=torch.einsum("jv,vck->jck", bm.J_regressor, bm.shapedirs).contiguous(),
                         parents=bm.parents, lbs_weights=bm.lbs_weights, shapedirs=bm.shapedirs, posedirs=bm.posedirs,
                         # bit j: the vertex has a skinning weight on joint j (anr_frame_backward skips what a joint cannot move)
                         vjmask=((bm.lbs_weights != 0).to(torch.int64) << torch.arange(bm.lbs_weights.shape[1], device=dev)).sum(1).to(torch.int32)
                         if bm.lbs_weights.shape[1] <= 31 else None)
            self._chain_const_cache = c
        return dict(c, T_template=self.verts_transform_template.detach().contiguous())

    def _attach_chain(self, rays_body, rays_world, o2c):
        """(rays_body, o2c) with the per-frame chain's backward attached (either may be None): ONE anr_frame_backward launch
        serves both gradients."""
        from .autograd import FrameChainFunction
        p = self._refine
        bs = (rays_body if rays_body is not None else o2c).shape[0]
        dev = (rays_body if rays_body is not None else o2c).device
        get = lambda k, n: p[k] if p.get(k) is not None else torch.zeros(bs, n, device=dev)
        betas, go, bp, transl = get("betas", 10), get("global_orient", 3), get("body_pose", 69), get("transl", 3)
        with torch.no_grad():                                   # the values the forward kernels consumed, packed once
            packed = (betas.detach().expand(bs, -1).contiguous(), torch.cat([go.detach(), bp.detach()], 1),
                      transl.detach().expand(bs, -1).contiguous())
        return FrameChainFunction.apply(betas, go, bp, transl, rays_body, o2c, self._chain_consts(), rays_world, packed)

    def convert_to_body_model_space(self, rays):
        """rays[bs,R,>=8] -> rays in the root-joint frame; moves the cached body state too."""
        if not self._pose_grad() and rays.is_cuda:
            # one launch for G^-1 and the body state, one for the rays (no LAPACK inverse: that call synchronises the host)
            g_inv, self.global_transform, self.verts, self.joints, self.verts_transform = ops.to_root_frame(
                self.global_transform, self.verts, self.joints, self.verts_transform)
            self._knn_index = None
            new_rays = ops.rays_to_body(g_inv, rays)
            if self._refine is not None:
                # the chain's two consumed outputs leave through one autograd node: ober2cano is computed here already
                o2c = self._ober2cano_values() if getattr(self, "verts_transform_template", None) is not None else None
                new_rays, o2c = self._attach_chain(new_rays, rays.detach(), o2c)
                self._o2c_attached = (self.verts_transform, o2c)
            return new_rays
        # tensor-op form: CPU, or gradients that must flow through torch autograd
        g_inv = _affine_inverse(self.global_transform) if self._pose_grad() else torch.inverse(self.global_transform)
        if self._pose_grad():                                              # per-frame, differentiable form of the kernel
            o = batch_transform(g_inv[:, None], rays[..., 0:3])
            d = batch_transform(g_inv[:, None], rays[..., 3:6], pad_ones=False)
            dist = torch.norm(o, dim=-1, keepdim=True)
            new_rays = torch.cat([o, d, torch.max(rays[..., 6:7], dist - 1.0), torch.min(rays[..., 7:8], dist + 1.0)], -1)
        else:
            new_rays = ops.rays_to_body(g_inv, rays)
        G = g_inv[:, None]
        self.verts = batch_transform(G, self.verts)
        self._knn_index = None
        self.joints = batch_transform(G, self.joints)
        self.global_transform = g_inv @ self.global_transform
        self.verts_transform = (small_matmul(G, self.verts_transform) if self._pose_grad()
                                else G @ self.verts_transform)
        return new_rays

    def clac_ober2cano_transform(self):
        if self._pose_grad():
            self.ober2cano_transform = _ober2cano_autograd(
                self.verts_transform, self.verts_transform_template,
                (self.shape_offsets_template - self.shape_offsets) + (self.pose_offsets_template - self.pose_offsets))
            return
        if self._refine is not None:
            hit = self._o2c_attached
            if hit is not None and hit[0] is self.verts_transform and hit[1] is not None:
                self.ober2cano_transform = hit[1]
            else:
                self.ober2cano_transform = self._attach_chain(None, None, self._ober2cano_values())[1]
            return
        self.ober2cano_transform = self._ober2cano_values()

    def _ober2cano_values(self):
        return ops.ober2cano(self.verts_transform, self.verts_transform_template, self.shape_offsets,
                             self.shape_offsets_template, self.pose_offsets, self.pose_offsets_template)

    # ------------------------------------------------------------------ per-point queries
    def _net(self, use_fine):
        return self.nerf_fine if use_fine else self.nerf

    def knn_index(self):
        """Spatial index over the current posed vertices (rebuilt when set_body_model /
        convert_to_body_model_space replace them)."""
        if self._knn_index is None or self._knn_index[0] is not self.verts:
            # (with the reach mask for this model's validity radius: ANR_WARP_NO_REACH_MASK=1 builds the plain index — A/B, tests)
            reach = 0.0 if os.environ.get("ANR_WARP_NO_REACH_MASK") else float(self.dis_threshold)
            self._knn_index = (self.verts, ops.knn_index_build(self.verts.detach(), self.knn_order, reach=reach))
        return self._knn_index[1]

    def _warp_generic(self, xyz, want_transform=False, chunk=1 << 19):
        """models/anim_nerf.py:153-192 for k_neigh != 4: exact k-NN (anr_knn_k, distances carry no gradient — the KNN_CUDA
        branch, :157-159), confidence + blend + transform as tensor ops (differentiable w.r.t. ober2cano and xyz).
        xyz[bs,N,>=3] -> pts[bs,N,4] = (canonical xyz, valid) (+ the blended transforms [bs,N,4,4])."""
        bs, V = self.ober2cano_transform.shape[:2]
        lbs, flat = self.body_model.lbs_weights, self.ober2cano_transform.reshape(bs * V, 4, 4)
        off = (torch.arange(bs, device=xyz.device) * V)[:, None, None]
        pts, Ts = [], []
        for s in range(0, xyz.shape[1], chunk):
            x = xyz[:, s:s + chunk, :3]
            with torch.no_grad():
                dist, idx = ops.knn_k(self.verts.detach(), x.detach().contiguous(), self.k_neigh)
                w_n = lbs[idx]
                conf = torch.exp(-(w_n - w_n[..., 0:1, :]).abs().sum(-1) / (2.0 * self.weight_std ** 2))
                w = torch.exp(-dist) * (conf > 0.9).float()
                w = w / w.sum(-1, keepdim=True)
                valid = ((w * dist).sum(-1, keepdim=True) < self.dis_threshold).float()
            T = (w[..., None, None] * flat[idx + off]).sum(2)
            pts.append(torch.cat([batch_transform(T, x), valid], -1))
            if want_transform:
                Ts.append(T)
        pts = torch.cat(pts, 1)
        return (pts, torch.cat(Ts, 1)) if want_transform else pts

    def unpose(self, xyz, viewdir=None):
        """-> xyz_unposed[bs,N,3], viewdir, valid[bs,N,1]   (models/anim_nerf.py:180-192).  With use_view and unpose_view
        the directions go through the sample's blended transform too — as POINTS (batch_transform's default pad_ones=True
        at :188-190 adds the translation), exactly as the reference does."""
        carry = self.use_view and self.unpose_view and viewdir is not None
        if self.k_neigh != 4:
            pts, T = self._warp_generic(xyz, want_transform=True)
            if carry:
                viewdir = batch_transform(T, viewdir.reshape(pts.shape[0], -1, 3), pad_ones=True)
            return pts[..., :3], viewdir, pts[..., 3:4]
        res = ops.warp_points(self.knn_index(), self.ober2cano_transform.detach(), self.body_model.lbs_weights,
                              self.dis_threshold, xyz=xyz, neighbours=carry)
        if not carry:
            return res[..., :3], viewdir, res[..., 3:4]
        pts, nidx, nw = res
        bs, V = self.ober2cano_transform.shape[:2]
        with torch.no_grad():
            # blended transform of every sample from the kernel's neighbour ids / weights: sum_k w_k ober2cano[id_k]
            flat = self.ober2cano_transform.detach().reshape(bs * V, 16)
            gid = (nidx.long() + (torch.arange(bs, device=nidx.device) * V)[:, None, None]).reshape(-1)
            T = (flat[gid].view(bs, -1, 4, 16) * nw[..., None]).sum(2).view(bs, -1, 4, 4)
            viewdir = batch_transform(T, viewdir.reshape(bs, -1, 3), pad_ones=True)
        return pts[..., :3], viewdir, pts[..., 3:4]

    def query_canonical_space(self, xyz, viewdir=None, use_fine=False, only_sigma=False, only_normal=False):
        net = self._net(use_fine)
        if only_sigma:
            return net.get_sigma(xyz, only_sigma=True)
        if only_normal:
            return net.get_normal(xyz)
        return net(xyz, viewdir=viewdir)

    def warped_points(self, *, xyz=None, rays=None, z=None, skip_far=False, lean=False, reuse=None, keep=None, steps=None):
        """pts[bs*N,4] = (canonical xyz, valid) for explicit points or for samples along rays.
        skip_far: provably-invalid samples (farther than dis_threshold from the body's bounding box) skip the
        neighbour search; only legal where sigma = -1e5 is all that is consumed (the renderer).
        keep (a dict, training): the call leaves what a later call on a superset of its samples can copy instead of searching
        again under keep["train"] = (pts, nbr_idx, nbr_w); reuse = that tuple + (perm,): sorted sample j of a ray is sample
        perm[j] of the kept call where perm[j] < its K (the fine pass of a step, anr_warp_points_reuse)."""
        if self.use_unpose and self.k_neigh != 4:
            if lean:
                raise NotImplementedError("the lean renderer schedule is built for k_neigh = 4")
            if xyz is None:
                xyz = (rays[..., None, :3] + z[..., None] * rays[..., None, 3:6]).reshape(rays.shape[0], -1, 3)
            return self._warp_generic(xyz).view(-1, 4)
        if self.use_unpose:
            far = skip_far and self.skip_far_samples
            if xyz is None and torch.is_grad_enabled() and (rays.requires_grad or z.requires_grad
                                                            or self.ober2cano_transform.requires_grad):
                from .autograd import WarpFunction               # pose refinement: differentiable warp
                return WarpFunction.apply(rays, z, self.ober2cano_transform, self.knn_index(),
                                          self.body_model.lbs_weights, self.dis_threshold, far, reuse if far else None,
                                          keep if far else None).view(-1, 4)
            if lean:                                            # (pts, valid bytes, valid list, device count)
                # (steps: the coarse pass's deterministic depths as their step table — no depth array, ops.warp_points)
                pts, vm, vi, vc = ops.warp_points(self.knn_index(), self.ober2cano_transform.detach(),
                                                  self.body_model.lbs_weights, self.dis_threshold, xyz=xyz, rays=rays,
                                                  z=z, skip_far=True, lean=True, reuse=reuse, steps=steps)
                return pts.view(-1, 4), vm, vi, vc
            if far and xyz is None and reuse is not None:        # training without pose refinement: rows only
                reuse = (reuse[0], None, reuse[3])
            else:
                reuse = None
            pts = ops.warp_points(self.knn_index(), self.ober2cano_transform.detach(), self.body_model.lbs_weights,
                                  self.dis_threshold, xyz=xyz, rays=rays, z=z, skip_far=far, reuse=reuse)
            if keep is not None and far and xyz is None:
                keep["train"] = (pts, None, None)
            return pts.view(-1, 4)
        if xyz is not None:
            flat = xyz.reshape(-1, xyz.shape[-1])[:, :3]
            return torch.cat([flat, torch.ones_like(flat[:, :1])], -1)
        return ops.points_from_rays(rays, z)

    def forward(self, xyz, viewdir=None, use_fine=False):
        """xyz[bs,nv,3] -> rgb[bs,nv,3], sigma[bs,nv,1]; sigma = -1e5 outside dis_threshold."""
        bs, nv = xyz.shape[:2]
        if self.use_view and self.unpose_view and self.use_unpose and viewdir is not None:
            xyz_c, viewdir, valid = self.unpose(xyz, viewdir)
            pts = torch.cat([xyz_c, valid], -1).view(-1, 4)
        else:
            pts = self.warped_points(xyz=xyz)
        if self.use_view:                       # view-dependent colour: the head runs outside the fused kernels
            out = self._net(use_fine).eval_points_view(pts, viewdir).view(bs, nv, 4).clone()
            out[..., 3] = torch.where(pts[:, 3].view(bs, nv) < 1, torch.full_like(out[..., 3], -1e5), out[..., 3])
            return out[..., :3], out[..., 3:4]
        out = self._net(use_fine).eval_points(pts, only_valid=self.use_unpose and self.query_inside).view(bs, nv, 4)
        return out[..., :3], out[..., 3:4]
