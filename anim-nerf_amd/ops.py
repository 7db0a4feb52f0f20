"""Tensor-level wrappers around the C ABI: one Python function per `anr_*` entry point.

PyTorch is only plumbing here (device memory, the current HIP stream).  Every function
requires CUDA(=HIP) float32 tensors and raises otherwise — there is no eager fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (ANR_MLP_BF16, ANR_MLP_F32, ANR_MLP_FLAG_NO_DMA, ANR_MLP_FLAG_SIGMA_ONLY, ANR_MLP_FLAG_TANGENT,
                   AnrMlpParams)

ANR_MLP_FLAG_W4 = 0x200
MLP_MODES = {"f32": ANR_MLP_F32, "fp32": ANR_MLP_F32, "bf16": ANR_MLP_BF16,
             # A/B variants of the same arithmetic (bench/diagnostics)
             "bf16_w4": ANR_MLP_BF16 | ANR_MLP_FLAG_W4, "bf16_nodma": ANR_MLP_BF16 | ANR_MLP_FLAG_NO_DMA,
             "f32_nodma": ANR_MLP_F32 | ANR_MLP_FLAG_NO_DMA}

# When set to a list (bench.py does), every launch appends (name, start_event, end_event, units, bytes): HIP events
# recorded on the stream the kernel is launched on, so elapsed_time() is that kernel's device time.  `bytes` = what the
# launch moves through HBM by design (each input read once, each output written once), for the HBM-bound kernels.
KERNEL_TIMING = None


class _timed:
    def __init__(self, name, units, nbytes=None):
        self.name, self.units, self.nbytes = name, units, nbytes

    def __enter__(self):
        self.on = KERNEL_TIMING is not None
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            KERNEL_TIMING.append((self.name, self.e0, self.e1, self.units, self.nbytes))
        return False


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the GPU (the HIP rendering path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        # the library launches on the calling thread's current device (hipLaunchKernelGGL, hipGetDevice): a tensor of
        # another GPU would be touched through a foreign stream
        raise RuntimeError(f"{name} lives on cuda:{t.device.index} but the current device is cuda:{torch.cuda.current_device()}: "
                           "call torch.cuda.set_device (one process per GPU) or wrap the call in torch.cuda.device(...)")
    # A strided input comes back as a contiguous COPY.  The caller must bind it to a local that lives until its launch is
    # enqueued (never `_ptr(_dev(x))`): a copy nobody holds is freed at once, the caching allocator may hand the block to
    # the next such copy, and that copy's kernel is enqueued BEFORE the launch — stream order only protects against
    # allocations made after it.
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


# ---------------------------------------------------------------------------------------------

def ray_gen(c2w: torch.Tensor, H: int, W: int, focal: torch.Tensor, near: float, far: float,
            center: Optional[torch.Tensor] = None) -> torch.Tensor:
    """rays[H,W,8] — datasets/anim_nerf_dataset.py:72-85."""
    lib = _lib.load()
    c2w = _dev(c2w, "c2w")
    focal = _dev(focal, "focal")
    if center is None:
        center = torch.tensor([W * 0.5, H * 0.5], dtype=torch.float32, device=c2w.device)
    center = _dev(center, "center")
    rays = torch.empty(H, W, 8, dtype=torch.float32, device=c2w.device)
    _lib.check(lib.anr_ray_gen(_ptr(c2w), _ptr(focal), _ptr(center), H, W, float(near), float(far),
                               _ptr(rays), _stream(rays)), "anr_ray_gen")
    return rays


def smpl_forward(betas, pose, transl, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    """SMPL forward / LBS (smplx/body_models.py:289-387, smplx/lbs.py:152-404) -> verts, joints[bs,J,3], A, T,
    shape_offsets, pose_offsets.  No autograd: the training path with pose refinement uses the tensor-op form."""
    lib = _lib.load()
    betas, pose, transl = _dev(betas, "betas"), _dev(pose, "pose"), _dev(transl, "transl")
    v_template, shapedirs, posedirs = _dev(v_template, "v_template"), _dev(shapedirs, "shapedirs"), _dev(posedirs, "posedirs")
    J_regressor, lbs_weights = _dev(J_regressor, "J_regressor"), _dev(lbs_weights, "lbs_weights")
    parents = _dev(parents, "parents", torch.int64)
    bs, V, J, NB = pose.shape[0], v_template.shape[0], J_regressor.shape[0], betas.shape[1]
    dev = pose.device
    new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    verts, joints, A, T = new(bs, V, 3), new(bs, J, 3), new(bs, J, 4, 4), new(bs, V, 4, 4)
    so, po = new(bs, V, 3), new(bs, V, 3)
    ws1, ws2, ws3 = new(bs, V, 3), new(bs, (V + 255) // 256, J, 3), new(bs, 9 * (J - 1))
    _lib.check(lib.anr_smpl_forward(_ptr(betas), _ptr(pose), _ptr(transl), bs, NB, _ptr(v_template), _ptr(shapedirs),
                                    _ptr(posedirs), _ptr(J_regressor), _ptr(parents), _ptr(lbs_weights), V, J,
                                    _ptr(verts), _ptr(joints), _ptr(A), _ptr(T), _ptr(so), _ptr(po), _ptr(ws1), _ptr(ws2),
                                    _ptr(ws3), _stream(verts)), "anr_smpl_forward")
    return verts, joints, A, T, so, po


def frame_backward(betas, pose, transl, J0, JS, parents, lbs_weights, shapedirs, posedirs, T_template, rays_world=None,
                   d_o2c=None, d_rays=None, vertex_joint_mask=None, forward_mode: bool = False, chain_values=None,
                   workspace=None) -> torch.Tensor:
    """dL/d(betas | global_orient | body_pose | transl)[bs,85] of the per-frame chain (SMPL/LBS, root frame, ober2cano) from
    dL/d ober2cano[bs,V,4,4] and / or dL/d rays_body[bs,R,8] (csrc/frame_bwd.hip).  forward_mode=True: the one-launch
    forward-mode kernel (one workgroup per frame and parameter), kept as the cross-check of the adjoint kernels."""
    lib = _lib.load()
    betas, pose, transl = _dev(betas, "betas"), _dev(pose, "pose"), _dev(transl, "transl")
    bs, V = pose.shape[0], lbs_weights.shape[0]
    T_template = _dev(T_template, "T_template")
    R, rs = 0, 0
    if d_rays is not None:
        d_rays, rays_world = _dev(d_rays, "d_rays"), _dev(rays_world, "rays_world")
        R, rs = rays_world.shape[1], rays_world.shape[2]
    if d_o2c is not None:
        d_o2c = _dev(d_o2c, "d_o2c")
    grads = torch.empty(bs, 85, dtype=torch.float32, device=pose.device)
    J0, JS, parents = _dev(J0, "J0"), _dev(JS, "JS"), _dev(parents, "parents", torch.int64)
    lbs_weights, shapedirs, posedirs = _dev(lbs_weights, "lbs_weights"), _dev(shapedirs, "shapedirs"), _dev(posedirs, "posedirs")
    if vertex_joint_mask is not None:
        vertex_joint_mask = _dev(vertex_joint_mask, "vertex_joint_mask", torch.int32)
    if not forward_mode:
        # reverse mode through the per-vertex inverses, forward mode through the 24-joint chain (csrc/frame_bwd.hip)
        # chain_values = (joints_transform[bs,J,4,4], g_inv[bs,4,4]) of the forward pass: the per-vertex kernel reads them instead
        # of walking the 24-joint chain again in every workgroup
        A_fwd, g_inv = (None, None) if chain_values is None else (_dev(chain_values[0], "joints_transform"), _dev(chain_values[1], "g_inv"))
        # workspace = frame_backward_workspace(bs, V) whose accumulators the caller has zeroed (zero_segments): no fill here
        ws = workspace if workspace is not None else torch.empty(lib.anr_frame_backward_ws_floats(bs, V), dtype=torch.float32, device=pose.device)
        with _timed("frame_backward", bs):
            _lib.check(lib.anr_frame_backward_adjoint_values(
                _ptr(betas), _ptr(pose), _ptr(transl), bs, _ptr(J0), _ptr(JS), _ptr(parents), _ptr(lbs_weights), _ptr(shapedirs),
                _ptr(posedirs), V, _ptr(T_template), T_template.shape[0], _ptr(rays_world), rs, R, _ptr(d_o2c),
                _ptr(d_rays), _ptr(A_fwd), _ptr(g_inv), _ptr(ws), _ptr(grads), 1 if workspace is not None else 0, _stream(grads)),
                "anr_frame_backward_adjoint")
        return grads
    with _timed("frame_backward", bs):
        _lib.check(lib.anr_frame_backward(_ptr(betas), _ptr(pose), _ptr(transl), bs, _ptr(J0), _ptr(JS), _ptr(parents),
                                          _ptr(lbs_weights), _ptr(shapedirs), _ptr(posedirs), V,
                                          _ptr(T_template), T_template.shape[0], _ptr(rays_world), rs, R, _ptr(d_o2c),
                                          _ptr(d_rays), _ptr(vertex_joint_mask),
                                          _ptr(grads), _stream(grads)), "anr_frame_backward")
    return grads


def frame_setup(tables, frame_idx, consts, bm, template, rays_world=None):
    """The per-frame set-up of a batch in two launches (anr_frame_setup): `tables` = (betas, global_orient, body_pose, transl)
    — BodyModelParams' tables with frame_idx[bs] int64, or per-frame arrays with frame_idx None; consts = {"J0", "JS"};
    bm the SMPL module's buffers; template = (T[bs_t,V,4,4], shape_off[bs_t,V,3], pose_off[bs_t,V,3]); rays_world[bs,R,8].
    -> dict(betas, pose, transl, A, joints, g_inv, g_root, shape_offsets, pose_offsets, verts, verts_transform (root frame),
    ober2cano, rays_body)."""
    lib = _lib.load()
    bw, gw, pw, tw = (_dev(t, n) for t, n in zip(tables, ("betas", "global_orient", "body_pose", "transl")))
    if frame_idx is not None:
        frame_idx = _dev(frame_idx, "frame_idx", torch.int64)
        bs = frame_idx.numel()
    else:
        bs = gw.shape[0]
        if bw.shape[0] != bs or pw.shape[0] != bs or tw.shape[0] != bs:
            raise ValueError("frame_setup: per-frame arrays of one batch size")
    dev = gw.device
    V, J, NB = bm.lbs_weights.shape[0], bm.lbs_weights.shape[1], bm.shapedirs.shape[-1]
    Tt, sot, pot = (_dev(t, n) for t, n in zip(template, ("T_template", "shape_offsets_template", "pose_offsets_template")))
    R, rs = 0, 0
    if rays_world is not None:
        rays_world = _dev(rays_world, "rays_world")
        R, rs = rays_world.shape[1], rays_world.shape[2]
    f32 = dict(dtype=torch.float32, device=dev)
    o = dict(betas=torch.empty(bs, NB, **f32), pose=torch.empty(bs, 3 * J, **f32), transl=torch.empty(bs, 3, **f32),
             A=torch.empty(bs, J, 4, 4, **f32), joints=torch.empty(bs, J, 3, **f32), g_inv=torch.empty(bs, 4, 4, **f32),
             g_root=torch.empty(bs, 4, 4, **f32), shape_offsets=torch.empty(bs, V, 3, **f32), pose_offsets=torch.empty(bs, V, 3, **f32),
             verts=torch.empty(bs, V, 3, **f32), verts_transform=torch.empty(bs, V, 4, 4, **f32), ober2cano=torch.empty(bs, V, 4, 4, **f32),
             rays_body=torch.empty(bs, R, 8, **f32) if R else None)
    ws = torch.empty(bs, 9 * (J - 1), **f32)
    J0, JS = _dev(consts["J0"], "J0"), _dev(consts["JS"], "JS")
    # (the pose tables' row count travels along: a frame index outside them reads row 0 and poisons the frame with NaN instead
    # of reading out of bounds — nn.Embedding raises there, models/body_model_params.py:5-68)
    _lib.check(lib.anr_frame_setup_rows(
        _ptr(frame_idx), gw.shape[0] if frame_idx is not None else 0, _ptr(bw), bw.shape[0], _ptr(gw), _ptr(pw), _ptr(tw), bs, _ptr(J0), _ptr(JS), _ptr(_dev(bm.parents, "parents", torch.int64)),
        _ptr(_dev(bm.v_template, "v_template")), _ptr(_dev(bm.shapedirs, "shapedirs")), _ptr(_dev(bm.posedirs, "posedirs")),
        _ptr(_dev(bm.lbs_weights, "lbs_weights")), V, J, NB, _ptr(Tt), _ptr(sot), _ptr(pot), Tt.shape[0], _ptr(rays_world), rs, R,
        _ptr(o["betas"]), _ptr(o["pose"]), _ptr(o["transl"]), _ptr(o["A"]), _ptr(o["joints"]), _ptr(o["g_inv"]), _ptr(o["g_root"]),
        _ptr(o["shape_offsets"]), _ptr(o["pose_offsets"]), _ptr(o["verts"]), _ptr(o["verts_transform"]), _ptr(o["ober2cano"]),
        _ptr(o["rays_body"]), _ptr(ws), _stream(ws)), "anr_frame_setup")
    return o


def to_root_frame(global_transform, verts, joints, T):
    """models/anim_nerf.py:128-144 for the body state: -> (g_inv[bs,4,4], G_root[bs,4,4], verts, joints, T) in the
    root-joint frame, one launch (closed-form affine inverse: no LAPACK call, no host synchronisation)."""
    lib = _lib.load()
    G, verts, joints, T = _dev(global_transform, "global_transform"), _dev(verts, "verts"), _dev(joints, "joints"), _dev(T, "T")
    bs, V, J = verts.shape[0], verts.shape[1], joints.shape[1]
    g_inv, g_root = torch.empty_like(G), torch.empty_like(G)
    v2, j2, T2 = torch.empty_like(verts), torch.empty_like(joints), torch.empty_like(T)
    _lib.check(lib.anr_to_root_frame(_ptr(G), _ptr(verts), _ptr(joints), _ptr(T), bs, V, J, _ptr(g_inv), _ptr(g_root), _ptr(v2),
                                     _ptr(j2), _ptr(T2), _stream(T2)), "anr_to_root_frame")
    return g_inv, g_root, v2, j2, T2


def rays_to_body(g_inv: torch.Tensor, rays: torch.Tensor) -> torch.Tensor:
    """models/anim_nerf.py:128-137.  g_inv[bs,4,4], rays[bs,R,>=8] -> [bs,R,8]."""
    lib = _lib.load()
    g_inv = _dev(g_inv, "g_inv")
    rays = _dev(rays, "rays")
    bs, R, stride = rays.shape
    out = torch.empty(bs, R, 8, dtype=torch.float32, device=rays.device)
    _lib.check(lib.anr_rays_to_body(_ptr(g_inv), _ptr(rays), _ptr(out), bs, R, stride, _stream(rays)),
               "anr_rays_to_body")
    return out


def ober2cano(t_pose, t_template, shape_off, shape_off_t, pose_off, pose_off_t) -> torch.Tensor:
    """models/anim_nerf.py:147-151.  t_pose[bs,V,4,4], shape_off / pose_off[bs,V,3]; the template's [bs or 1, V, ...] (one
    template for all frames is read in place, not expanded) -> [bs,V,4,4]."""
    lib = _lib.load()
    t_pose = _dev(t_pose, "t_pose")
    t_template = _dev(t_template, "t_template")
    so, sot = _dev(shape_off, "shape_off"), _dev(shape_off_t, "shape_off_t")
    po, pot = _dev(pose_off, "pose_off"), _dev(pose_off_t, "pose_off_t")
    tb = t_template.shape[0]
    if tb not in (1, t_pose.shape[0]) or sot.shape[0] != tb or pot.shape[0] != tb:
        raise ValueError("template batch must be 1 or the frames' batch")
    out = torch.empty_like(t_pose)
    n = t_pose.shape[0] * t_pose.shape[1]
    _lib.check(lib.anr_ober2cano(_ptr(t_pose), _ptr(t_template), _ptr(so), _ptr(sot), _ptr(po), _ptr(pot),
                                 _ptr(out), n, tb * t_pose.shape[1], _stream(out)), "anr_ober2cano")
    return out


def morton_order(points: torch.Tensor) -> torch.Tensor:
    """int32 permutation (slot -> vertex id) that sorts points[V,3] along a 30-bit Morton curve.
    Host-side, once per model: gives anr_knn_index_build clusters of spatial neighbours."""
    p = points.detach().float().cpu()
    lo, hi = p.min(0).values, p.max(0).values
    q = ((p - lo) / (hi - lo).clamp_min(1e-12) * 1023.0).long().clamp_(0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(code, stable=True).to(torch.int32)


def knn_index_build(verts: torch.Tensor, order: Optional[torch.Tensor] = None, reach: float = 0.0) -> torch.Tensor:
    """Per-frame spatial index of posed vertices verts[bs,V,3] -> uint8[bs, anr_knn_index_bytes(V)].
    reach > 0: with the reach mask for validity radii <= reach (anr_knn_index_build_reach): warp_points(skip_far) drops the
    samples no vertex can reach in its classify pass."""
    lib = _lib.load()
    verts = _dev(verts, "verts")
    bs, V, _ = verts.shape
    if order is not None:
        order = _dev(order, "order", torch.int32)
    nbytes = lib.anr_knn_index_bytes(V)
    index = torch.empty(bs, nbytes, dtype=torch.uint8, device=verts.device)
    _lib.check(lib.anr_knn_index_build_reach(_ptr(verts), _ptr(order), bs, V, float(reach), _ptr(index), _stream(index)),
               "anr_knn_index_build")
    return index


def knn(verts: torch.Tensor, xyz: torch.Tensor, index: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Drop-in for knn_cuda.KNN(k=4, transpose_mode=True)(ref, query) — models/anim_nerf.py:159."""
    lib = _lib.load()
    verts, xyz = _dev(verts, "verts"), _dev(xyz, "xyz")
    bs, V, _ = verts.shape
    N = xyz.shape[1]
    if index is None:
        index = knn_index_build(verts, morton_order(verts[0]).to(verts.device))
    dist = torch.empty(bs, N, 4, dtype=torch.float32, device=xyz.device)
    idx = torch.empty(bs, N, 4, dtype=torch.int64, device=xyz.device)
    with _timed("knn", bs * N):
        _lib.check(lib.anr_knn(_ptr(index), _ptr(xyz), bs, V, N, _ptr(dist), _ptr(idx), _stream(xyz)), "anr_knn")
    return dist, idx


def knn_k(verts: torch.Tensor, xyz: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """knn_cuda.KNN(k, transpose_mode=True)(ref, query) for any k in 1..8 (exhaustive exact search; k = 4 has the pruned
    search of `knn`).  verts[bs,V,3], xyz[bs,N,>=3] -> dist[bs,N,k] ascending, idx[bs,N,k] int64."""
    lib = _lib.load()
    verts, xyz = _dev(verts, "verts"), _dev(xyz, "xyz")
    bs, V, _ = verts.shape
    N = xyz.shape[1]
    dist = torch.empty(bs, N, k, dtype=torch.float32, device=xyz.device)
    idx = torch.empty(bs, N, k, dtype=torch.int64, device=xyz.device)
    with _timed("knn_k", bs * N):
        _lib.check(lib.anr_knn_k(_ptr(verts), _ptr(xyz), xyz.shape[2], bs, V, N, int(k), _ptr(dist), _ptr(idx), _stream(xyz)),
                   "anr_knn_k")
    return dist, idx


def sample_coarse(rays: torch.Tensor, steps: torch.Tensor, t_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """models/volume_rendering.py:29-56.  rays[..., >=8] (flattened to R) -> z[R,K]."""
    lib = _lib.load()
    rays = _dev(rays, "rays")
    steps = _dev(steps, "steps")
    stride = rays.shape[-1]
    R = rays.numel() // stride
    K = steps.numel()
    z = torch.empty(R, K, dtype=torch.float32, device=rays.device)
    if t_rand is not None:
        t_rand = _dev(t_rand, "t_rand")
    with _timed("sample_coarse", R * K, R * (8 + 4 * K)):
        _lib.check(lib.anr_sample_coarse(_ptr(rays), stride, _ptr(steps), _ptr(t_rand), R, K, _ptr(z), _stream(z)),
                   "anr_sample_coarse")
    return z


def frame_backward_workspace(bs: int, V: int, device):
    """-> (workspace for frame_backward(workspace=), its leading slice the caller must zero before the call)"""
    lib = _lib.load()
    ws = torch.empty(lib.anr_frame_backward_ws_floats(bs, V), dtype=torch.float32, device=device)
    return ws, ws[:lib.anr_frame_backward_ws_zero_floats(bs)]


def warp_workspace(bs: int, N: int, device):
    """-> (workspace for warp_points(workspace=), the slice of it the caller must zero before the call)"""
    import ctypes as C
    lib = _lib.load()
    ws = torch.empty(lib.anr_warp_ws_ints(bs, N), dtype=torch.int32, device=device)
    first, n = C.c_int64(0), C.c_int64(0)
    _lib.check(lib.anr_warp_ws_zero_range(bs, N, C.byref(first), C.byref(n)), "anr_warp_ws_zero_range")
    return ws, ws[first.value:first.value + n.value]


def warp_points(index, o2c, lbs_weights, dis_threshold: float, *, xyz=None, rays=None, z=None, debug=False,
                skip_far=False, neighbours=False, two_pass=True, lean=False, reuse=None, workspace=None, steps=None):
    """models/anim_nerf.py:153-192.  Either xyz[bs,N,3|4] or (rays[bs,R,>=8], z[bs,R,K]).
    `index` = knn_index_build(posed verts).  Returns pts[bs,N,4] = (x_c, y_c, z_c, valid)
    (+ dist[bs,N,4], idx[bs,N,4] i32, blended[bs,N] if debug).
    steps[K] instead of z (lean renderer pass only): the step table of the deterministic stratified depths,
    z_k = near' (1 - s_k) + far' s_k computed in the kernel (the bits of sample_coarse) — no depth array."""
    lib = _lib.load()
    index = _dev(index, "knn_index", torch.uint8)
    o2c, lbs_weights = _dev(o2c, "ober2cano"), _dev(lbs_weights, "lbs_weights")
    bs, V = o2c.shape[0], o2c.shape[1]
    J = lbs_weights.shape[1]
    if debug and skip_far:
        raise ValueError("debug outputs need skip_far=False")
    if xyz is not None:
        xyz = _dev(xyz, "xyz")
        N, xs = xyz.shape[1], xyz.shape[2]
        rs, K = 0, 0
    elif steps is not None:
        if z is not None or not (lean and skip_far and two_pass) or reuse is not None or neighbours:
            raise ValueError("warp_points(steps=): the lean renderer pass of the coarse samples (no z, no reuse, no neighbour outputs)")
        rays, z = _dev(rays, "rays"), _dev(steps, "steps")
        K = z.numel()
        N = rays.shape[1] * K
        rs, xs = rays.shape[-1], 0
    else:
        rays, z = _dev(rays, "rays"), _dev(z, "z")
        K = z.shape[-1]
        N = z.shape[1] * K
        rs, xs = rays.shape[-1], 0
    dev = o2c.device
    pts = torch.empty(bs, N, 4, dtype=torch.float32, device=dev)
    dist = idx = blended = None
    if debug:
        dist = torch.empty(bs, N, 4, dtype=torch.float32, device=dev)
        idx = torch.empty(bs, N, 4, dtype=torch.int32, device=dev)
        blended = torch.empty(bs, N, dtype=torch.float32, device=dev)
    nidx = nw = None
    if neighbours:                       # training: vertex ids + blend weights of the 4 neighbours (zeros where skipped)
        nidx = torch.empty(bs, N, 4, dtype=torch.int32, device=dev)
        nw = torch.empty(bs, N, 4, dtype=torch.float32, device=dev)
    # renderer mode: classify + compact the samples near the body, then search the compacted list
    # (workspace = warp_workspace(bs, N)[0] with its counters zeroed by the caller: the call launches no fill of its own)
    ws = ((workspace if workspace is not None else torch.empty(lib.anr_warp_ws_ints(bs, N), dtype=torch.int32, device=dev))
          if (skip_far and two_pass) else None)
    vmask = vindex = vcount = None
    if lean:
        # renderer only: validity as one byte per sample + the list of valid positions; pts rows of far samples stay
        # unwritten (-> (pts, valid[bs,N] u8, index[bs*N] i32, count[1] i32 on the device))
        if not (skip_far and two_pass):
            raise ValueError("lean=True needs skip_far=True and two_pass=True")
        vmask = torch.empty(bs, N, dtype=torch.uint8, device=dev)
        vindex = torch.empty(bs * N, dtype=torch.int32, device=dev)
        vcount = torch.empty(1, dtype=torch.int32, device=dev)
    r_pts = r_mask = r_perm = r_nidx = r_nw = None
    r_k = 0
    if reuse is not None:
        # inference (lean): (pts, valid bytes, perm) of the coarse call; training: (pts, None, perm, nbr_idx, nbr_w) — see the header
        if xyz is not None or not (skip_far and two_pass):
            raise ValueError("reuse needs rays mode, skip_far=True and two_pass=True")
        if lean != (reuse[1] is not None):
            raise ValueError("reuse: the coarse call's validity bytes go with lean=True, and only with it")
        r_pts, r_perm = _dev(reuse[0], "reuse pts"), _dev(reuse[2], "perm", torch.uint8)
        r_mask = None if reuse[1] is None else _dev(reuse[1], "reuse mask", torch.uint8)
        r_k = r_pts.numel() // 4 // (bs * (N // K))
        if neighbours:
            r_nidx, r_nw = _dev(reuse[3], "reuse nbr_idx", torch.int32), _dev(reuse[4], "reuse nbr_w")
    # (lean renderer: the coarse call's workspace rides on its validity bytes — `warp_cells` below — and the fine call that
    # reuses those bytes copies the per-cell search results out of it instead of searching the cells again: anr_warp_points_cells)
    prev_ws, prev_n = None, 0
    if lean and r_mask is not None and dis_threshold == getattr(reuse[1], "warp_cells", (None, 0, None, None))[2] \
            and reuse[1].warp_cells[3] == index.data_ptr():
        prev_ws, prev_n = reuse[1].warp_cells[0], reuse[1].warp_cells[1]
    with _timed("warp_points", bs * N):
        _lib.check(lib.anr_warp_points_cells(_ptr(xyz), xs, _ptr(rays), rs, _ptr(z), K, _ptr(index), _ptr(o2c),
                                             _ptr(lbs_weights), bs, V, J, N, float(dis_threshold),
                                             ((3 if (workspace is not None and ws is not None) else 1) | (4 if steps is not None else 0)) if skip_far else 0,
                                             _ptr(pts), _ptr(dist), _ptr(idx), _ptr(blended), _ptr(nidx), _ptr(nw), _ptr(ws),
                                             _ptr(vmask), _ptr(vindex), _ptr(vcount), _ptr(r_pts), _ptr(r_mask), _ptr(r_perm),
                                             r_k, _ptr(r_nidx), _ptr(r_nw), _ptr(prev_ws), prev_n, _stream(pts)),
                   "anr_warp_points")
    if lean:
        vmask.warp_cells = (ws, N, dis_threshold, index.data_ptr())       # (keeps the workspace alive as long as the bytes)
        return pts, vmask, vindex, vcount
    if neighbours:
        return pts, nidx, nw
    return (pts, dist, idx, blended) if debug else pts


def warp_backward(d_pts, rays, z, o2c, nbr_idx, nbr_w):
    """Backward of warp_points(rays=, z=): -> d_o2c[bs,V,4,4], d_rays[bs,R,8], d_z[bs,R,K]."""
    lib = _lib.load()
    d_pts, rays, z, o2c = _dev(d_pts, "d_pts"), _dev(rays, "rays"), _dev(z, "z"), _dev(o2c, "ober2cano")
    nbr_idx, nbr_w = _dev(nbr_idx, "nbr_idx", torch.int32), _dev(nbr_w, "nbr_w")
    bs, R, K = z.shape
    V = o2c.shape[1]
    d_o2c = torch.zeros_like(o2c)
    d_rays = torch.zeros(bs, R, 8, dtype=torch.float32, device=z.device)
    d_z = torch.empty_like(z)
    with _timed("warp_backward", bs * R * K):
        _lib.check(lib.anr_warp_backward(_ptr(d_pts), _ptr(rays), rays.shape[-1], _ptr(z), K, _ptr(o2c), _ptr(nbr_idx),
                                         _ptr(nbr_w), bs, V, R * K, _ptr(d_o2c), _ptr(d_rays), _ptr(d_z), _stream(z)),
                   "anr_warp_backward")
    return d_o2c, d_rays, d_z


def points_from_rays(rays: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """x = o + z d, valid = 1 (use_unpose=False).  rays[..,R,>=8], z[..,R,K] -> pts[R*K,4]."""
    lib = _lib.load()
    rays, z = _dev(rays, "rays"), _dev(z, "z")
    K = z.shape[-1]
    n = z.numel()
    pts = torch.empty(n, 4, dtype=torch.float32, device=z.device)
    with _timed("points_from_rays", n, n * 20 + (n // K) * 32):
        _lib.check(lib.anr_points_from_rays(_ptr(rays), rays.shape[-1], _ptr(z), K, n, _ptr(pts), _stream(pts)),
                   "anr_points_from_rays")
    return pts


def mlp_pack_pair(params_a: dict, params_b: dict, mode: int, backward: bool = False):
    """mlp_pack of two networks of the plain architecture in ONE launch (anr_mlp_pack_pair / anr_mlp_bwd_pack_pair)."""
    lib = _lib.load()
    keep, sts = [], []
    for params in (params_a, params_b):
        st = AnrMlpParams()

        def g(key):
            t = _dev(params[key].detach(), key)
            keep.append(t)
            return t.data_ptr()
        for i in range(8):
            st.w_trunk[i] = g(f"xyz_encoding_{i+1}.0.weight")
            st.b_trunk[i] = g(f"xyz_encoding_{i+1}.0.bias")
        st.w_sigma, st.b_sigma = g("sigma.weight"), g("sigma.bias")
        st.w_final, st.b_final = g("xyz_encoding_final.weight"), g("xyz_encoding_final.bias")
        st.w_dir, st.b_dir = g("dir_encoding.0.weight"), g("dir_encoding.0.bias")
        st.w_rgb, st.b_rgb = g("rgb.0.weight"), g("rgb.0.bias")
        shapes = [(256, 63)] + [(256, 256)] * 3 + [(256, 319)] + [(256, 256)] * 3
        for i, shp in enumerate(shapes):
            if tuple(params[f"xyz_encoding_{i+1}.0.weight"].shape) != shp:
                raise ValueError(f"xyz_encoding_{i+1}.0.weight: expected {shp}")
        if tuple(params["dir_encoding.0.weight"].shape) != (128, 256) or tuple(params["rgb.0.weight"].shape) != (3, 128):
            raise ValueError("head shapes must be dir_encoding [128,256], rgb [3,128] (no latent codes)")
        sts.append(st)
    nbytes = lib.anr_mlp_bwd_pack_bytes(mode & 0xff) if backward else lib.anr_mlp_pack_bytes(mode & 0xff)
    pa, pb = (torch.empty(nbytes, dtype=torch.uint8, device=keep[0].device) for _ in range(2))
    fn = lib.anr_mlp_bwd_pack_pair if backward else lib.anr_mlp_pack_pair
    _lib.check(fn(C.byref(sts[0]), C.byref(sts[1]), mode & 0xff, _ptr(pa), _ptr(pb), _stream(pa)), "anr_mlp_pack_pair")
    return pa, pb


def mlp_pack(params: dict, mode: int, backward: bool = False, view_channels: int = 0) -> torch.Tensor:
    """Pack the 11 weight/bias tensors (reference state-dict keys, PyTorch [out,in] layout) for anr_mlp_forward, or —
    `backward` — their transposes for anr_mlp_backward.  view_channels = 3 + 6 freqs_dir: the pack of mlp_forward_view
    (use_view=True: dir_encoding.0.weight is [128, 256 + view_channels])."""
    lib = _lib.load()
    keep = []

    def g(key):
        t = _dev(params[key].detach(), key)
        keep.append(t)
        return t.data_ptr()
    st = AnrMlpParams()
    for i in range(8):
        st.w_trunk[i] = g(f"xyz_encoding_{i+1}.0.weight")
        st.b_trunk[i] = g(f"xyz_encoding_{i+1}.0.bias")
    st.w_sigma, st.b_sigma = g("sigma.weight"), g("sigma.bias")
    st.w_final, st.b_final = g("xyz_encoding_final.weight"), g("xyz_encoding_final.bias")
    st.w_dir, st.b_dir = g("dir_encoding.0.weight"), g("dir_encoding.0.bias")
    st.w_rgb, st.b_rgb = g("rgb.0.weight"), g("rgb.0.bias")
    shapes = [(256, 63)] + [(256, 256)] * 3 + [(256, 319)] + [(256, 256)] * 3
    for i, shp in enumerate(shapes):
        if tuple(params[f"xyz_encoding_{i+1}.0.weight"].shape) != shp:
            raise ValueError(f"xyz_encoding_{i+1}.0.weight: expected {shp}")
    if tuple(params["dir_encoding.0.weight"].shape) != (128, 256 + view_channels) or tuple(params["rgb.0.weight"].shape) != (3, 128):
        raise ValueError(f"head shapes must be dir_encoding [128,{256 + view_channels}], rgb [3,128] (no latent codes)")
    if view_channels:
        if backward:
            raise ValueError("the view-dependent head is fused in inference only")
        pack = torch.empty(lib.anr_mlp_pack_bytes((mode & 0xff) | _lib.ANR_MLP_FLAG_VIEW), dtype=torch.uint8, device=keep[0].device)
        _lib.check(lib.anr_mlp_pack_view(C.byref(st), mode & 0xff, view_channels, _ptr(pack), _stream(pack)), "anr_mlp_pack_view")
        return pack
    if backward:
        pack = torch.empty(lib.anr_mlp_bwd_pack_bytes(mode & 0xff), dtype=torch.uint8, device=keep[0].device)
        _lib.check(lib.anr_mlp_bwd_pack(C.byref(st), mode & 0xff, _ptr(pack), _stream(pack)), "anr_mlp_bwd_pack")
        return pack
    nbytes = lib.anr_mlp_pack_bytes(mode & 0xff)
    pack = torch.empty(nbytes, dtype=torch.uint8, device=keep[0].device)
    _lib.check(lib.anr_mlp_pack(C.byref(st), mode & 0xff, _ptr(pack), _stream(pack)), "anr_mlp_pack")
    return pack


def encode(pts: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """enc[n,63] = Embedding(pts[:, :3]) (models/embedding.py:22-39) in one launch, fp32 or bf16."""
    lib = _lib.load()
    pts = _dev(pts, "pts")
    n = pts.shape[0]
    enc = torch.empty(n, 63, dtype=dtype, device=pts.device)
    _lib.check(lib.anr_encode(_ptr(pts), pts.shape[1], n, 1 if dtype == torch.bfloat16 else 0, _ptr(enc), _stream(enc)),
               "anr_encode")
    return enc


def encode64(pts: torch.Tensor, dtype=torch.float32, tangent: bool = False, count: Optional[torch.Tensor] = None) -> torch.Tensor:
    """enc[n,64] = [Embedding(pts[:, :3]), 0]: the operand layout of mlp_wgrad.  tangent: rows in quads, rows 4p+1..3 =
    d enc / d x, y, z of point p (forward-mode normals).  count: device int32 — only that many rows are computed."""
    lib = _lib.load()
    pts = _dev(pts, "pts")
    n = pts.shape[0]
    enc = torch.empty(n, 64, dtype=dtype, device=pts.device)
    flags = (1 if dtype == torch.bfloat16 else 0) | (ANR_MLP_FLAG_TANGENT if tangent else 0)
    if count is not None:
        _lib.check(lib.anr_encode64_counted(_ptr(pts), pts.shape[1], n, _ptr(count), flags, _ptr(enc), _stream(enc)), "anr_encode64")
    else:
        _lib.check(lib.anr_encode64(_ptr(pts), pts.shape[1], n, flags, _ptr(enc), _stream(enc)), "anr_encode64")
    return enc


_WGRAD_WS = {}


def mlp_wgrad(mode: int, act: torch.Tensor, dact: torch.Tensor, enc: torch.Tensor, g4: torch.Tensor, sigma_only: bool = False,
              tangent: bool = False, accumulate_into: Optional[torch.Tensor] = None, count: Optional[torch.Tensor] = None,
              background: bool = False, out: Optional[torch.Tensor] = None, no_fill: bool = False):
    """All 22 parameter gradients of one network from the saved activations, the activation gradients, the encoding matrix
    (encode64) and g4 = (dL/d rgb_pre, dL/d sigma): a flat fp32 tensor in the order of autograd.PARAM_KEYS (PyTorch
    [out][in] layouts).  Hand-written split-K MFMA GEMMs (csrc/mlp_wgrad.hip); n % 64 == 0."""
    lib = _lib.load()
    act, dact, enc, g4 = _dev(act, "act", act.dtype), _dev(dact, "dact", act.dtype), _dev(enc, "enc", act.dtype), _dev(g4, "g4")
    n = act.shape[0]
    m = (mode & 0xff) | (ANR_MLP_FLAG_SIGMA_ONLY if sigma_only else 0) | (ANR_MLP_FLAG_TANGENT if tangent else 0)
    if background:                                         # runs next to other launches: half the slices (ANR_MLP_FLAG_BACKGROUND)
        m |= _lib.ANR_MLP_FLAG_BACKGROUND
    key = (act.device.index, torch.cuda.current_stream(act.device).cuda_stream)
    ws = _WGRAD_WS.get(key)
    need = lib.anr_mlp_wgrad_ws_floats(n)
    if ws is None or ws.numel() < need:                    # per-stream scratch, reused by every call (split-K partials)
        ws = _WGRAD_WS[key] = torch.empty(need, dtype=torch.float32, device=act.device)
    if accumulate_into is not None:                        # += into a flat gradient buffer (autograd.GradSink)
        grads = _dev(accumulate_into, "accumulate_into")
        if grads.numel() != lib.anr_mlp_wgrad_floats():
            raise ValueError("accumulate_into must hold anr_mlp_wgrad_floats() floats")
        m |= _lib.ANR_MLP_FLAG_ACCUMULATE
    else:
        grads = out if out is not None else torch.empty(lib.anr_mlp_wgrad_floats(), dtype=torch.float32, device=act.device)
        if grads.numel() != lib.anr_mlp_wgrad_floats() or not grads.is_contiguous():
            raise ValueError("out must hold anr_mlp_wgrad_floats() floats")
        if no_fill:                                        # sigma_only: floats behind anr_mlp_wgrad_sigma_floats() are left alone
            m |= _lib.ANR_MLP_FLAG_NO_FILL
    with _timed("mlp_wgrad", n if count is None else count):
        if count is not None:                              # rows on the device (a multiple of 64 <= n); n = the buffers' rows
            _lib.check(lib.anr_mlp_wgrad_counted(m, _ptr(act), _ptr(dact), _ptr(enc), _ptr(g4), n, _ptr(count), _ptr(ws), _ptr(grads),
                                                 _stream(grads)), "anr_mlp_wgrad")
        else:
            _lib.check(lib.anr_mlp_wgrad(m, _ptr(act), _ptr(dact), _ptr(enc), _ptr(g4), n, _ptr(ws), _ptr(grads), _stream(grads)),
                       "anr_mlp_wgrad")
    return grads


def mlp_denc(mode: int, dact: torch.Tensor, w1: torch.Tensor, w5: torch.Tensor, count: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d_enc[n,63] = dact_1 . W1[:, :63] + dact_5 . W5[:, :63] (fp32): the gradient leaving the MLP through its encoding
    inputs (pose refinement).  w1 / w5 = xyz_encoding_1 / _5 weights as stored (fp32)."""
    lib = _lib.load()
    dact = _dev(dact, "dact", dact.dtype)
    w1, w5 = _dev(w1.detach(), "w1"), _dev(w5.detach(), "w5")
    n = dact.shape[0]
    d_enc = torch.empty(n, 63, dtype=torch.float32, device=dact.device)
    with _timed("mlp_denc", n if count is None else count):
        if count is not None:
            _lib.check(lib.anr_mlp_denc_counted(mode & 0xff, _ptr(dact), _ptr(w1), _ptr(w5), n, _ptr(count), _ptr(d_enc), _stream(d_enc)),
                       "anr_mlp_denc")
        else:
            _lib.check(lib.anr_mlp_denc(mode & 0xff, _ptr(dact), _ptr(w1), _ptr(w5), n, _ptr(d_enc), _stream(d_enc)), "anr_mlp_denc")
    return d_enc


def mlp_dpoints(bwd_pack: torch.Tensor, mode: int, dact: torch.Tensor, pts: torch.Tensor, count: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d_pts[n,4] = (dL/dxyz, 0) of the rows of a compacted pass: `mlp_denc` + `encode_backward` in one launch, the encoding
    panels of layers 1 and 5 read from the backward weight pack (anr_mlp_dpoints)."""
    lib = _lib.load()
    dact, pts = _dev(dact, "dact", dact.dtype), _dev(pts, "pts")
    n = dact.shape[0]
    d_pts = torch.empty(n, 4, dtype=torch.float32, device=dact.device)
    with _timed("mlp_dpoints", n if count is None else count):
        _lib.check(lib.anr_mlp_dpoints(_ptr(bwd_pack), mode & 0xff, _ptr(dact), _ptr(pts), n, _ptr(count), _ptr(d_pts), _stream(d_pts)),
                   "anr_mlp_dpoints")
    return d_pts


def encode_backward(pts: torch.Tensor, d_enc: torch.Tensor, count: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d_pts[n,4] = (dL/dxyz, 0) from d_enc[n,63] (fp32)."""
    lib = _lib.load()
    pts, d_enc = _dev(pts, "pts"), _dev(d_enc, "d_enc")
    n = pts.shape[0]
    d_pts = torch.empty(n, 4, dtype=torch.float32, device=pts.device)
    if count is not None:
        _lib.check(lib.anr_encode_backward_counted(_ptr(pts), pts.shape[1], _ptr(d_enc), n, _ptr(count), _ptr(d_pts), _stream(d_pts)),
                   "anr_encode_backward")
    else:
        _lib.check(lib.anr_encode_backward(_ptr(pts), pts.shape[1], _ptr(d_enc), n, _ptr(d_pts), _stream(d_pts)),
                   "anr_encode_backward")
    return d_pts


ACT_COLS = 2432                 # columns of a saved-activation / activation-gradient buffer: h1..h8 | final | dir hidden


def act_columns(act: torch.Tensor, c0: int = 0, c1: int = ACT_COLS) -> torch.Tensor:
    """Columns [c0, c1) (multiples of 32) of a saved-activation or activation-gradient buffer as a row-major [n, c1 - c0]
    tensor (a copy).  The kernels keep these buffers BLOCKED by 32-column tile and by 16-byte piece of a row inside it — 76 x
    (4 bf16 / 8 fp32) arrays of [n][16 bytes], then the ReLU sign bits (csrc/mlp_core.h) — inside the n x anr_mlp_act_cols() elements `mlp_forward_save` / `mlp_backward` return."""
    n = act.shape[0]
    assert c0 % 32 == 0 and c1 % 32 == 0 and 0 <= c0 < c1 <= ACT_COLS
    epp = 16 // act.element_size()                               # inside a block: piece arrays [n][16 bytes] (8 bf16 / 4 fp32)
    pieces = act.reshape(-1)[:ACT_COLS * n].view(ACT_COLS // epp, n, epp)
    return pieces[c0 // epp:c1 // epp].permute(1, 0, 2).reshape(n, c1 - c0)


def mlp_backward(bwd_pack: torch.Tensor, mode: int, g: torch.Tensor, act: torch.Tensor, sigma_only: bool = False,
                 tangent: bool = False, count: Optional[torch.Tensor] = None, enc_only: bool = False):
    """g[n,4] = (dL/d rgb_pre, dL/d sigma), act from mlp_forward_save -> dact (same dtype, same blocked layout: see
    `act_columns`): the pre-activation gradient of every layer (trunk columns only if sigma_only).
    enc_only: frozen networks — only the gradients of layers 1 and 5 (what mlp_denc reads) are written (ANR_MLP_FLAG_ENC_ONLY)."""
    lib = _lib.load()
    g, act = _dev(g, "g"), _dev(act, "act", act.dtype)
    n = g.shape[0]
    dact = torch.empty_like(act)
    m = ((mode & 0xff) | (ANR_MLP_FLAG_SIGMA_ONLY if sigma_only else 0) | (ANR_MLP_FLAG_TANGENT if tangent else 0)
         | (_lib.ANR_MLP_FLAG_ENC_ONLY if enc_only else 0))
    with _timed("mlp_backward_enc" if enc_only else "mlp_backward", n if count is None else count):
        if count is not None:
            _lib.check(lib.anr_mlp_backward_counted(_ptr(bwd_pack), m, _ptr(g), _ptr(act), _ptr(dact), n, _ptr(count), _stream(dact)),
                       "anr_mlp_backward")
        else:
            _lib.check(lib.anr_mlp_backward(_ptr(bwd_pack), m, _ptr(g), _ptr(act), _ptr(dact), n, _stream(dact)),
                       "anr_mlp_backward")
    return dact


def mlp_backward_feature(bwd_pack: torch.Tensor, mode: int, g: torch.Tensor, d_feature: torch.Tensor, act: torch.Tensor) -> torch.Tensor:
    """mlp_backward entered at xyz_encoding_final: g[n,4] (4th column = dL/d sigma), d_feature[n,256] fp32 -> dact."""
    lib = _lib.load()
    g, d_feature, act = _dev(g, "g"), _dev(d_feature, "d_feature"), _dev(act, "act", act.dtype)
    n = act.shape[0]
    dact = torch.empty_like(act)
    with _timed("mlp_backward", n):
        _lib.check(lib.anr_mlp_backward_feature(_ptr(bwd_pack), mode & 0xff, _ptr(g), _ptr(d_feature), _ptr(act), _ptr(dact), n,
                                                _stream(dact)), "anr_mlp_backward_feature")
    return dact


def compact_valid(pts: torch.Tensor, fill: Optional[torch.Tensor] = None):
    """-> (index[n] int32, count[1] int32 on the device): positions of the samples with valid >= 1
    (`inside_inds`, models/anim_nerf.py:253).  fill[n,4] or fill[n]: rows of the other samples := (0,0,0,-1e5) / -1e5."""
    lib = _lib.load()
    pts = _dev(pts, "pts")
    n = pts.numel() // 4
    index = torch.empty(n, dtype=torch.int32, device=pts.device)
    count = torch.empty(1, dtype=torch.int32, device=pts.device)
    cols = 0 if fill is None else (4 if fill.dim() == 2 else 1)
    with _timed("compact_valid", n, n * 20):
        _lib.check(lib.anr_compact_valid(_ptr(pts), n, _ptr(index), _ptr(count), _ptr(fill), cols, _stream(pts)),
                   "anr_compact_valid")
    return index, count


def mlp_forward(pack: torch.Tensor, mode: int, pts: torch.Tensor, sigma_only: bool = False,
                only_valid: bool = False, valid_list=None) -> torch.Tensor:
    """pts[n,4] = (x,y,z,valid) -> out[n,4] = (r,g,b,sigma), or sigma[n] if sigma_only.
    only_valid: evaluate the samples with valid >= 1 only; the others get (0,0,0,-1e5) as in the reference's
    query_canonical_space_inside (models/anim_nerf.py:245-290).  Sigma is the same either way; rgb of an invalid
    sample differs (0 instead of the colour of a point that is never composited)."""
    lib = _lib.load()
    pts = _dev(pts, "pts")
    n = pts.numel() // 4
    if sigma_only:
        mode = (mode & ~0x300) | ANR_MLP_FLAG_SIGMA_ONLY
        out = torch.empty(n, dtype=torch.float32, device=pts.device)
    else:
        out = torch.empty(n, 4, dtype=torch.float32, device=pts.device)
    if valid_list is not None:
        # (index, count) from warp_points(lean=True): rows of the other samples stay uninitialised — their consumer
        # (composite(valid=...)) does not read them
        index, count = valid_list
        with _timed("mlp_forward", count):
            _lib.check(lib.anr_mlp_forward_indexed(_ptr(pack), mode, _ptr(pts), _ptr(index), _ptr(count), n, _ptr(out),
                                                   _stream(out)), "anr_mlp_forward_indexed")
        return out
    if only_valid:
        index, count = compact_valid(pts, fill=out)
        with _timed("mlp_forward", count):                 # units resolved after the timed region (device counter)
            _lib.check(lib.anr_mlp_forward_indexed(_ptr(pack), mode, _ptr(pts), _ptr(index), _ptr(count), n, _ptr(out),
                                                   _stream(out)), "anr_mlp_forward_indexed")
        return out
    with _timed("mlp_forward", n):
        _lib.check(lib.anr_mlp_forward(_ptr(pack), mode, _ptr(pts), n, _ptr(out), _stream(out)), "anr_mlp_forward")
    return out


def mlp_forward_view(pack: torch.Tensor, mode: int, pts: torch.Tensor, viewdir: torch.Tensor, only_valid: bool = False) -> torch.Tensor:
    """use_view=True, inference: out[n,4] = (r,g,b,sigma) of pts[n,4] = (x,y,z,valid) seen along viewdir[n,3], the whole
    network — view-dependent colour head included — in the fused kernel (pack from mlp_pack(view_channels=...)).
    only_valid: as mlp_forward."""
    lib = _lib.load()
    pts, viewdir = _dev(pts, "pts"), _dev(viewdir, "viewdir")
    n = pts.numel() // 4
    if viewdir.numel() != 3 * n:
        raise ValueError("one view direction per point")
    out = torch.empty(n, 4, dtype=torch.float32, device=pts.device)
    index = count = None
    if only_valid:
        index, count = compact_valid(pts, fill=out)
    with _timed("mlp_forward", n if count is None else count):
        _lib.check(lib.anr_mlp_forward_view(_ptr(pack), mode & 0xff, _ptr(pts), _ptr(viewdir), 3, _ptr(index), _ptr(count), n, _ptr(out),
                                            _stream(out)), "anr_mlp_forward_view")
    return out


def mlp_forward_rays(pack: torch.Tensor, mode: int, rays: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """use_unpose=False: out[n,4] = NeRF(o + z d) for rays[bs,R,>=8], z[bs,R,K], n = bs R K; the sample points are
    generated inside the MLP kernel (no pts array)."""
    lib = _lib.load()
    rays, z = _dev(rays, "rays"), _dev(z, "z")
    K = z.shape[-1]
    n = z.numel()
    out = torch.empty(n, 4, dtype=torch.float32, device=z.device)
    with _timed("mlp_forward", n):
        _lib.check(lib.anr_mlp_forward_rays(_ptr(pack), mode & 0xff, _ptr(rays), rays.shape[-1], _ptr(z), K, n, _ptr(out),
                                            _stream(out)), "anr_mlp_forward_rays")
    return out


def mlp_forward_embedded(pack: torch.Tensor, mode: int, emb: torch.Tensor, sigma_only: bool = False, want_act: bool = False):
    """emb[n,63] (already Fourier-embedded, models/mlp.py:268-297) -> out[n,4] = (r,g,b,sigma), or sigma[n];
    want_act: (out, act[n, anr_mlp_act_cols()]) as mlp_forward_save."""
    lib = _lib.load()
    emb = _dev(emb, "emb")
    if emb.shape[-1] != 63:
        raise ValueError("mlp_forward_embedded: expected 63 embedded channels (in_channels_xyz = 3 + 3*10*2)")
    n = emb.numel() // 63
    mode = (mode & 0xff) | (ANR_MLP_FLAG_SIGMA_ONLY if sigma_only else 0)
    out = torch.empty((n,) if sigma_only else (n, 4), dtype=torch.float32, device=emb.device)
    act = None
    if want_act:
        act = torch.empty(n, lib.anr_mlp_act_cols(), device=emb.device,
                          dtype=torch.bfloat16 if (mode & 0xff) == ANR_MLP_BF16 else torch.float32)
    with _timed("mlp_forward", n):
        _lib.check(lib.anr_mlp_forward_embedded(_ptr(pack), mode, _ptr(emb), n, _ptr(out), _ptr(act), _stream(out)),
                   "anr_mlp_forward_embedded")
    return (out, act) if want_act else out


def mlp_forward_rays_steps(pack: torch.Tensor, mode: int, rays: torch.Tensor, steps: torch.Tensor) -> torch.Tensor:
    """As mlp_forward_rays with the deterministic stratified depths z = near' (1 - steps) + far' steps computed in the
    kernel too: no depth array.  rays[bs,R,>=8], steps[K] -> out[bs*R*K, 4]."""
    lib = _lib.load()
    rays, steps = _dev(rays, "rays"), _dev(steps, "steps")
    K = steps.numel()
    n = (rays.numel() // rays.shape[-1]) * K
    out = torch.empty(n, 4, dtype=torch.float32, device=rays.device)
    with _timed("mlp_forward", n):
        _lib.check(lib.anr_mlp_forward_rays_steps(_ptr(pack), mode & 0xff, _ptr(rays), rays.shape[-1], _ptr(steps), K, n,
                                                  _ptr(out), _stream(out)), "anr_mlp_forward_rays_steps")
    return out


def grid_points(N: int, x_range, y_range, z_range, center: torch.Tensor, first: int, count: int) -> torch.Tensor:
    """extract_mesh.py:27-35,152-157: slab [first, first+count) of the flattened N^3 grid -> pts[count,4]."""
    lib = _lib.load()
    center = _dev(center.reshape(-1), "center")
    pts = torch.empty(count, 4, dtype=torch.float32, device=center.device)
    _lib.check(lib.anr_grid_points(N, float(x_range[0]), float(x_range[1]), float(y_range[0]), float(y_range[1]),
                                   float(z_range[0]), float(z_range[1]), _ptr(center), first, count, _ptr(pts),
                                   _stream(pts)), "anr_grid_points")
    return pts


def composite(rgbs, z, rays, white_bkgd: bool, noise=None, want_weights: bool = True, valid=None, pos=None):
    """models/volume_rendering.py:131-160.  rgbs[R,K,4], z[R,K], rays[R,>=8].
    valid[R,K] uint8 (from warp_points(lean=True)): samples with 0 count as (0,0,0,-1e5) and their rows are not read.
    pos[R*K] int32 (training): rgbs = the rows of a compacted pass, looked up through pos (-1: an invalid sample, as above)."""
    lib = _lib.load()
    rgbs, z, rays = _dev(rgbs, "rgbs"), _dev(z, "z"), _dev(rays, "rays")
    R, K = z.shape
    if pos is not None:
        if valid is not None:
            raise ValueError("composite: pos or valid, not both")
        pos = _dev(pos, "pos", torch.int32)
        if pos.numel() < R * K:
            raise ValueError("composite: pos holds fewer than R K entries")
    dev = z.device
    w = torch.empty(R, K, dtype=torch.float32, device=dev) if want_weights else None
    rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(R, 1, dtype=torch.float32, device=dev)
    acc = torch.empty(R, 1, dtype=torch.float32, device=dev)
    if noise is not None:
        noise = _dev(noise, "noise")
    if valid is not None:
        valid = _dev(valid, "valid", torch.uint8)
    moved = R * (K * (20 if valid is None else 5) + 8 + 20 + (4 * K if want_weights else 0))
    if valid is not None:
        moved = None                  # rows of invalid samples are skipped: data-dependent, counted by the PMC passes only
    if pos is not None:
        with _timed("composite", R * K, None):
            _lib.check(lib.anr_composite_indexed(_ptr(rgbs), _ptr(pos), _ptr(z), _ptr(rays), rays.shape[-1], _ptr(noise), R, K,
                                                 1 if white_bkgd else 0, _ptr(w), _ptr(rgb), _ptr(depth), _ptr(acc), _stream(z)),
                       "anr_composite_indexed")
        return w, rgb, depth, acc
    with _timed("composite", R * K, moved):
        _lib.check(lib.anr_composite_masked(_ptr(rgbs), _ptr(z), _ptr(rays), rays.shape[-1], _ptr(noise), _ptr(valid), R, K,
                                            1 if white_bkgd else 0, _ptr(w), _ptr(rgb), _ptr(depth), _ptr(acc), _stream(z)),
                   "anr_composite")
    return w, rgb, depth, acc


def composite_backward(rgbs, z, rays, white_bkgd: bool, g_rgb, g_depth, g_acc, noise=None, g_weights=None,
                       want_dz: bool = False, out: Optional[torch.Tensor] = None, pos=None, g4_out: Optional[torch.Tensor] = None,
                       count: Optional[torch.Tensor] = None):
    """Backward of `composite`: -> d_rgbs[R,K,4] (and d_z[R,K], d_far[R] if want_dz).  out: a caller-owned buffer whose first
    R K rows receive d_rgbs (the explicit training step keeps rider rows behind them).  pos: as in `composite`.
    g4_out[rows,4] + count (with pos; anr_composite_backward_compact): the gradient goes straight to the g operand of the
    network's backward — row pos[sample] = (dL/d rgb . sigmoid', dL/d sigma) — nothing per sample is written, d_rgbs is None."""
    lib = _lib.load()
    rgbs, z, rays = _dev(rgbs, "rgbs"), _dev(z, "z"), _dev(rays, "rays")
    g_rgb, g_depth, g_acc = (None if g is None else _dev(g, nm) for g, nm in ((g_rgb, "g_rgb"), (g_depth, "g_depth"), (g_acc, "g_acc")))
    R, K = z.shape
    if noise is not None:
        noise = _dev(noise, "noise")
    if g_weights is not None:
        g_weights = _dev(g_weights, "g_weights")
    if out is not None:
        out = _dev(out, "out")
        if out.numel() < R * K * 4:
            raise ValueError("composite_backward: out is smaller than R K rows")
    dz = torch.empty(R, K, dtype=torch.float32, device=z.device) if want_dz else None
    dfar = torch.empty(R, dtype=torch.float32, device=z.device) if want_dz else None
    if g4_out is not None:
        if pos is None or count is None:
            raise ValueError("composite_backward: g4_out needs pos and count")
        g4_out, count = _dev(g4_out, "g4_out"), _dev(count, "count", torch.int32)
        with _timed("composite_backward", R * K):
            _lib.check(lib.anr_composite_backward_compact(_ptr(rgbs), _ptr(_dev(pos, "pos", torch.int32)), _ptr(count), _ptr(z), _ptr(rays),
                                                          rays.shape[-1], _ptr(noise), R, K, 1 if white_bkgd else 0, _ptr(g_weights),
                                                          _ptr(g_rgb), _ptr(g_depth), _ptr(g_acc), _ptr(g4_out), _ptr(dz), _ptr(dfar),
                                                          _stream(g4_out)), "anr_composite_backward_compact")
        return (None, dz, dfar) if want_dz else None
    d = out if out is not None else torch.empty(R, K, 4, dtype=torch.float32, device=z.device)
    with _timed("composite_backward", R * K):
        _lib.check(lib.anr_composite_backward_indexed(_ptr(rgbs), None if pos is None else _ptr(_dev(pos, "pos", torch.int32)), _ptr(z),
                                                      _ptr(rays), rays.shape[-1], _ptr(noise), R, K,
                                                      1 if white_bkgd else 0, _ptr(g_weights), _ptr(g_rgb), _ptr(g_depth),
                                                      _ptr(g_acc), _ptr(d), _ptr(dz), _ptr(dfar), _stream(d)),
                   "anr_composite_backward")
    return (d, dz, dfar) if want_dz else d


def mlp_forward_save(pack: torch.Tensor, mode: int, pts: torch.Tensor, sigma_only: bool = False, tangent: bool = False,
                     count: Optional[torch.Tensor] = None, bits_only: bool = False):
    """Training forward: (out, act) — act (n x anr_mlp_act_cols() elements, blocked by 32-column tile: `act_columns`) keeps
    every layer's post-activation output and the ReLU sign bits.
    tangent (with sigma_only): points in quads, rows 4p+1..3 carry d/dx, d/dy, d/dz (ANR_MLP_FLAG_TANGENT).
    bits_only: frozen networks (the `_refine` stage) — only the sign bits are written (ANR_MLP_FLAG_BITS_ONLY)."""
    lib = _lib.load()
    pts = _dev(pts, "pts")
    n = pts.numel() // 4
    mode = (mode & 0xff) | (_lib.ANR_MLP_FLAG_BITS_ONLY if bits_only else 0)
    if sigma_only:
        mode |= ANR_MLP_FLAG_SIGMA_ONLY | (ANR_MLP_FLAG_TANGENT if tangent else 0)
        out = torch.empty(n, dtype=torch.float32, device=pts.device)
    else:
        out = torch.empty(n, 4, dtype=torch.float32, device=pts.device)
    act = torch.empty(n, lib.anr_mlp_act_cols(), device=pts.device,
                      dtype=torch.bfloat16 if (mode & 0xff) == ANR_MLP_BF16 else torch.float32)
    with _timed("mlp_forward_bits" if bits_only else "mlp_forward_save", n if count is None else count):
        if count is not None:                              # rows on the device; act is sized (and blocked) for all n
            _lib.check(lib.anr_mlp_forward_save_indexed(_ptr(pack), mode, _ptr(pts), None, _ptr(count), n, _ptr(out), _ptr(act),
                                                        _stream(out)), "anr_mlp_forward_save")
        else:
            _lib.check(lib.anr_mlp_forward_save(_ptr(pack), mode, _ptr(pts), n, _ptr(out), _ptr(act), _stream(out)),
                       "anr_mlp_forward_save")
    return out, act


def sample_fine_merge(z_coarse, weights, u, want_fine: bool = False, want_perm: bool = False, perm_u8: bool = False):
    """models/volume_rendering.py:59-97,199-207.  z_coarse[R,Kc], weights[R,Kc], u[Kf] or u[R,Kf]."""
    lib = _lib.load()
    z_coarse, weights, u = _dev(z_coarse, "z_coarse"), _dev(weights, "weights"), _dev(u, "u")
    R, Kc = z_coarse.shape
    Kf = u.shape[-1]
    per_ray = 1 if u.dim() > 1 else 0
    dev = z_coarse.device
    zf = torch.empty(R, Kf, dtype=torch.float32, device=dev) if want_fine else None
    zs = torch.empty(R, Kc + Kf, dtype=torch.float32, device=dev)
    perm = torch.empty(R, Kc + Kf, dtype=torch.uint8 if perm_u8 else torch.int32, device=dev) if want_perm else None
    fn = lib.anr_sample_fine_merge_u8 if (want_perm and perm_u8) else lib.anr_sample_fine_merge
    with _timed("sample_fine_merge", R * (Kc + Kf), R * (8 * Kc + 4 * (Kc + Kf) + (4 * Kf if want_fine else 0)
                                                         + ((Kc + Kf) * (1 if perm_u8 else 4) if want_perm else 0))):
        _lib.check(fn(_ptr(z_coarse), _ptr(weights), _ptr(u), per_ray, R, Kc, Kf, _ptr(zf), _ptr(zs), _ptr(perm),
                      _stream(zs)), "anr_sample_fine_merge")
    if want_perm:
        return (zs, zf, perm) if want_fine else (zs, perm)
    return (zs, zf) if want_fine else zs


def composite_sample(rgbs, rays, u, white_bkgd: bool, *, z=None, steps=None, valid=None, want_weights=False,
                     want_fine=False, want_perm=False):
    """The coarse pass of an inference render in one launch: composite(rgbs[R,Kc,4]) + importance samples from its
    weights + merge (models/volume_rendering.py:172-178, :199-207).  Depths: z[R,Kc], or the step table steps[Kc] of the
    deterministic stratified depths (computed in the kernel).  -> dict(rgb, depth, acc, z_sorted[R,Kc+Kf], weights?,
    z_fine?, perm? (uint8))."""
    lib = _lib.load()
    rgbs, rays, u = _dev(rgbs, "rgbs"), _dev(rays, "rays"), _dev(u, "u")
    if (z is None) == (steps is None):
        raise ValueError("composite_sample: exactly one of z / steps")
    R, Kc = rgbs.shape[0], rgbs.shape[1]
    Kf = u.shape[-1]
    per_ray = 1 if u.dim() > 1 else 0
    dev = rgbs.device
    new = lambda *shape, dt=torch.float32: torch.empty(*shape, dtype=dt, device=dev)
    o = dict(rgb=new(R, 3), depth=new(R, 1), acc=new(R, 1), z_sorted=new(R, Kc + Kf))
    o["weights"] = new(R, Kc) if want_weights else None
    o["z_fine"] = new(R, Kf) if want_fine else None
    o["perm"] = new(R, Kc + Kf, dt=torch.uint8) if want_perm else None
    z = None if z is None else _dev(z, "z")
    steps = None if steps is None else _dev(steps, "steps")
    valid = None if valid is None else _dev(valid, "valid", torch.uint8)
    moved = R * (16 * Kc + (4 * Kc if z is not None else 0) + 8 + 20 + 4 * (Kc + Kf) + (4 * Kc if want_weights else 0)
                 + (4 * Kf if want_fine else 0) + ((Kc + Kf) if want_perm else 0))
    if valid is not None:
        moved = None
    with _timed("composite_sample", R * (Kc + Kc + Kf), moved):
        _lib.check(lib.anr_composite_sample(_ptr(rgbs), _ptr(z), _ptr(steps), _ptr(rays), rays.shape[-1], _ptr(valid), _ptr(u),
                                            per_ray, R, Kc, Kf, 1 if white_bkgd else 0, _ptr(o["weights"]), _ptr(o["rgb"]),
                                            _ptr(o["depth"]), _ptr(o["acc"]), _ptr(o["z_fine"]), _ptr(o["z_sorted"]),
                                            _ptr(o["perm"]), _stream(rgbs)), "anr_composite_sample")
    return o


def ray_march(pack_c: torch.Tensor, pack_f: torch.Tensor, mode: int, rays: torch.Tensor, steps: torch.Tensor, u: torch.Tensor,
              white_bkgd: bool, warp=None):
    """The whole coarse -> fine render of rays[R, >=8] in ONE launch (anr_ray_march: 64 + 64 samples)
    -> dict(rgb, depth, acc, rgb_fine, depth_fine, acc_fine), the bits of the staged launches.
    warp = (knn_index[bs, bytes], ober2cano[bs,V,4,4], lbs_weights[V,J], dis_threshold, rays_per_body): the inverse-LBS / 4-NN
    warp inside the pass (anr_ray_march_warp; rays in the bodies' root frames, R = bs * rays_per_body), dense evaluation."""
    lib = _lib.load()
    rays, steps, u = _dev(rays, "rays"), _dev(steps, "steps"), _dev(u, "u")
    R = rays.shape[0]
    new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=rays.device)
    o = dict(rgb=new(R, 3), depth=new(R, 1), acc=new(R, 1), rgb_fine=new(R, 3), depth_fine=new(R, 1), acc_fine=new(R, 1))
    outs = (_ptr(o["rgb"]), _ptr(o["depth"]), _ptr(o["acc"]), _ptr(o["rgb_fine"]), _ptr(o["depth_fine"]), _ptr(o["acc_fine"]))
    if warp is not None:
        index, o2c, lbs_w, thr, per_body = warp
        index, o2c, lbs_w = _dev(index, "knn_index", torch.uint8), _dev(o2c, "ober2cano"), _dev(lbs_w, "lbs_weights")
        bs, V = o2c.shape[0], o2c.shape[1]
        if R != bs * int(per_body):
            raise ValueError(f"ray_march(warp=): {R} rays for {bs} bodies x {per_body} rays")
        with _timed("ray_march_warp", R * (2 * steps.numel() + u.numel()), R * (32 + 40)):
            _lib.check(lib.anr_ray_march_warp(_ptr(pack_c), _ptr(pack_f), mode & 0xff, _ptr(rays), rays.shape[-1], bs, int(per_body),
                                              _ptr(steps), steps.numel(), _ptr(u), u.numel(), 1 if white_bkgd else 0, _ptr(index),
                                              _ptr(o2c), _ptr(lbs_w), V, lbs_w.shape[1], float(thr), *outs, _stream(rays)),
                       "anr_ray_march_warp")
        return o
    with _timed("ray_march", R * (2 * steps.numel() + u.numel()), R * (32 + 40)):
        _lib.check(lib.anr_ray_march(_ptr(pack_c), _ptr(pack_f), mode & 0xff, _ptr(rays), rays.shape[-1], R, _ptr(steps), steps.numel(),
                                     _ptr(u), u.numel(), 1 if white_bkgd else 0, *outs, _stream(rays)), "anr_ray_march")
    return o


# ---- the steps between the big kernels of a training step (csrc/train_glue.hip)
def compact_ordered(pts: torch.Tensor):
    """-> (index[n] int32 — first `count` entries: the positions with valid >= 1, ascending —, pos[n] int32 (row in that
    list or -1), pts_c[roundup64(n), 4] (the listed points, then zero rows to the next multiple of 64), count[2] int32 on
    the device: the listed rows, and that rounded up to a multiple of 64 — the row count of the *_counted kernels)."""
    lib = _lib.load()
    pts = _dev(pts, "pts")
    n = pts.numel() // 4
    index = torch.empty(n, dtype=torch.int32, device=pts.device)
    pos = torch.empty(n, dtype=torch.int32, device=pts.device)
    pts_c = torch.empty(-(-n // 64) * 64, 4, dtype=torch.float32, device=pts.device)
    count = torch.empty(2, dtype=torch.int32, device=pts.device)
    ws = torch.empty(lib.anr_compact_ws_ints(n), dtype=torch.int32, device=pts.device)
    with _timed("compact_ordered", n, n * 24):
        _lib.check(lib.anr_compact_ordered(_ptr(pts), n, _ptr(index), _ptr(pos), _ptr(pts_c), _ptr(count), _ptr(ws), _stream(pts)),
                   "anr_compact_ordered")
    return index, pos, pts_c, count


def expand_rows(src: torch.Tensor, pos: torch.Tensor, fill: float) -> torch.Tensor:
    """out[i] = src[pos[i]] where pos[i] >= 0, else (0,0,0,fill) (src[., 4]) or fill (src[.])."""
    lib = _lib.load()
    src = _dev(src, "src")
    n = pos.numel()
    cols = 4 if src.dim() == 2 else 1
    out = torch.empty((n, 4) if cols == 4 else (n,), dtype=torch.float32, device=src.device)
    _lib.check(lib.anr_expand_rows(_ptr(src), _ptr(pos), n, cols, float(fill), _ptr(out), _stream(out)), "anr_expand_rows")
    return out


def mlp_head_grad(g: torch.Tensor, index: Optional[torch.Tensor], out: Optional[torch.Tensor], pts: torch.Tensor, rows,
                  sigma_only: bool) -> torch.Tensor:
    """The g[n_pad,4] operand of mlp_backward / mlp_wgrad from the upstream gradient of (rgb, sigma) (see the header).
    rows: int, or the device count[2] of compact_ordered (rows past count[1] are left unwritten)."""
    lib = _lib.load()
    g, pts = _dev(g, "g"), _dev(pts, "pts")
    n_pad = pts.shape[0]
    g4 = torch.empty(n_pad, 4, dtype=torch.float32, device=pts.device)
    if isinstance(rows, torch.Tensor):
        _lib.check(lib.anr_mlp_head_grad_counted(_ptr(g), _ptr(index), _ptr(out), _ptr(pts), _ptr(rows), n_pad, 1 if sigma_only else 0,
                                                 _ptr(g4), _stream(g4)), "anr_mlp_head_grad")
    else:
        _lib.check(lib.anr_mlp_head_grad(_ptr(g), _ptr(index), _ptr(out), _ptr(pts), rows, n_pad, 1 if sigma_only else 0, _ptr(g4),
                                         _stream(g4)), "anr_mlp_head_grad")
    return g4


def tangent_quads(xyz: torch.Tensor, n_pad: int) -> torch.Tensor:
    """xyz[n,3] -> pts4[4 n_pad, 4]: four rows (x,y,z,1) per point, zero rows for the padding points."""
    lib = _lib.load()
    xyz = _dev(xyz, "xyz")
    pts4 = torch.empty(4 * n_pad, 4, dtype=torch.float32, device=xyz.device)
    _lib.check(lib.anr_tangent_quads(_ptr(xyz), xyz.shape[0], n_pad, _ptr(pts4), _stream(pts4)), "anr_tangent_quads")
    return pts4


def sample_coarse_backward(g_z: torch.Tensor, steps: torch.Tensor, t_rand: Optional[torch.Tensor]) -> torch.Tensor:
    """d_rays[R,8] (columns 6, 7 = d near', d far'; the rest zero) from dL/dz[R,K] of sample_coarse."""
    lib = _lib.load()
    g_z, steps = _dev(g_z, "g_z"), _dev(steps, "steps")
    K = steps.numel()
    R = g_z.numel() // K
    d_rays = torch.empty(R, 8, dtype=torch.float32, device=g_z.device)
    t_rand = None if t_rand is None else _dev(t_rand, "t_rand")
    _lib.check(lib.anr_sample_coarse_backward(_ptr(g_z), _ptr(steps), _ptr(t_rand), R, K,
                                              _ptr(d_rays), _stream(d_rays)), "anr_sample_coarse_backward")
    return d_rays


def merge_backward(g_sorted: torch.Tensor, perm: torch.Tensor, Kc: int) -> torch.Tensor:
    """dL/dz_coarse[R,Kc] from dL/dz_sorted[R,K] and sample_fine_merge's permutation (int32)."""
    lib = _lib.load()
    g_sorted, perm = _dev(g_sorted, "g_sorted"), _dev(perm, "perm", torch.int32)
    R, K = perm.shape
    d = torch.empty(R, Kc, dtype=torch.float32, device=g_sorted.device)
    _lib.check(lib.anr_merge_backward(_ptr(g_sorted), _ptr(perm), R, K, Kc, _ptr(d), _stream(d)), "anr_merge_backward")
    return d


_LOSS_WS = {}
LOSS_NAMES = ("loss_rgb", "loss_rgb_fine", "loss_alphas", "loss_alphas_fine", "loss_foreground", "loss_background",
              "loss_foreground_fine", "loss_background_fine", "loss_normals", "loss_normals_fine")


def _loss_args(t: dict, c: dict):
    a = _lib.AnrLossArgs()
    a._keep = []                                             # contiguous copies of strided inputs must outlive the launch
    for k in ("rgb", "acc", "rgb_fine", "acc_fine", "target_rgb", "target_alpha", "s", "s_fine", "quads", "quads_fine"):
        v = t.get(k)
        if v is not None:
            v = _dev(v, k)
            a._keep.append(v)
        setattr(a, k, None if v is None else _ptr(v))
    for k in ("R", "prior_rows", "nv", "normal_sets", "quad_rows", "n_fg", "n_bg"):
        setattr(a, k, int(c.get(k, 0)))
    for k in ("k", "delta", "lambda_alphas", "lambda_foreground", "lambda_background", "lambda_normals"):
        setattr(a, k, float(c.get(k, 0.0)))
    return a


def train_loss(tensors: dict, consts: dict) -> torch.Tensor:
    """vals[12]: the ten loss terms (LOSS_NAMES), their weighted total and the batch's PSNR, one launch (anr_train_loss)."""
    import ctypes as C
    lib = _lib.load()
    ref = next(v for v in tensors.values() if v is not None)
    key = (ref.device.index, torch.cuda.current_stream(ref.device).cuda_stream)
    ws = _LOSS_WS.get(key)
    if ws is None:
        ws = _LOSS_WS[key] = torch.zeros(lib.anr_train_loss_ws_floats(), dtype=torch.float32, device=ref.device)
    vals = torch.empty(12, dtype=torch.float32, device=ref.device)   # ten terms, the total, the batch's PSNR
    a = _loss_args(tensors, consts)
    _lib.check(lib.anr_train_loss(C.byref(a), _ptr(ws), _ptr(vals), _stream(vals)), "anr_train_loss")
    return vals


def train_loss_backward(tensors: dict, consts: dict, g_total: torch.Tensor, want: dict) -> dict:
    """Gradients of the total w.r.t. the tensors named in `want` (rgb, acc, rgb_fine, acc_fine, s, s_fine, quads, quads_fine)."""
    import ctypes as C
    lib = _lib.load()
    a = _loss_args(tensors, consts)
    d = _lib.AnrLossGrads()
    out = {}
    for k in ("rgb", "acc", "rgb_fine", "acc_fine", "s", "s_fine", "quads", "quads_fine"):
        if want.get(k) and tensors.get(k) is not None:
            out[k] = torch.empty_like(tensors[k], memory_format=torch.contiguous_format)
            setattr(d, k, _ptr(out[k]))
    g_total = _dev(g_total.reshape(1), "g_total")
    _lib.check(lib.anr_train_loss_backward(C.byref(a), _ptr(g_total), C.byref(d), _stream(g_total)), "anr_train_loss_backward")
    return out


_MC_TABLES = {}


def marching_cubes(volume: torch.Tensor, level: float = 0.0):
    """(vertices[V,3] float32 in index coordinates, triangles[T,3] int64) of the surface volume == level (inside = value <
    level; normals point out of the inside): marching cubes over volume[N0,N1,N2] in two launches (csrc/mesh.hip) with the
    case table of anim_nerf_amd.mesh.case_table."""
    from .mesh import case_table
    lib = _lib.load()
    volume = _dev(volume, "volume")
    n0, n1, n2 = volume.shape
    key = volume.device.index
    if key not in _MC_TABLES:
        n_tris, tris = case_table()
        _MC_TABLES[key] = (torch.from_numpy(n_tris).to(volume.device), torch.from_numpy(tris.reshape(-1)).to(volume.device))
    n_tris, tris = _MC_TABLES[key]
    total = n0 * n1 * n2
    vmask = torch.empty(total, dtype=torch.uint8, device=volume.device)
    vcnt = torch.empty(total, dtype=torch.int32, device=volume.device)
    tcnt = torch.empty(total, dtype=torch.int32, device=volume.device)
    with _timed("mc_classify", total, total * 13):
        _lib.check(lib.anr_mc_classify(_ptr(volume), n0, n1, n2, float(level), _ptr(n_tris), _ptr(vmask), _ptr(vcnt), _ptr(tcnt),
                                       _stream(volume)), "anr_mc_classify")
    vend, tend = torch.cumsum(vcnt, 0), torch.cumsum(tcnt, 0)               # int64
    V, T = int(vend[-1].item()), int(tend[-1].item())
    vstart, tstart = vend - vcnt, tend - tcnt
    verts = torch.empty(max(V, 1), 3, dtype=torch.float32, device=volume.device)
    faces = torch.empty(max(T, 1), 3, dtype=torch.int32, device=volume.device)
    with _timed("mc_emit", total, total * 21 + V * 12 + T * 12):
        _lib.check(lib.anr_mc_emit(_ptr(volume), n0, n1, n2, float(level), _ptr(n_tris), _ptr(tris), _ptr(vmask), _ptr(vstart),
                                   _ptr(tstart), _ptr(verts), _ptr(faces), _stream(volume)), "anr_mc_emit")
    return verts[:V], faces[:T].long()


def adam_step(chunks: torch.Tensor, n_chunks: int, step: torch.Tensor, lrs, beta1: float, beta2: float, eps: float,
              active: Optional[torch.Tensor] = None, ticket: Optional[torch.Tensor] = None) -> None:
    """torch.optim.Adam's update (train.py:216-226) over a table of tensor chunks in one launch (include/animnerf_hip.h:
    anr_adam_step); `step` is the device-side count of THIS update, `lrs` the host learning rates of the groups.
    active / ticket (int32[1], zero): `step` holds the counts BEFORE this update and the kernel advances them itself."""
    lib = _lib.load()
    chunks, step = _dev(chunks, "chunks", torch.uint8), _dev(step, "step")
    arr = (C.c_float * len(lrs))(*[float(x) for x in lrs])
    if active is not None:
        active, ticket = _dev(active, "active"), _dev(ticket, "ticket", torch.int32)
        _lib.check(lib.anr_adam_step_counting(_ptr(chunks), int(n_chunks), _ptr(step), _ptr(active), step.numel(), _ptr(ticket), arr, len(lrs),
                                              float(beta1), float(beta2), float(eps), _stream(step)), "anr_adam_step_counting")
        return
    _lib.check(lib.anr_adam_step(_ptr(chunks), int(n_chunks), _ptr(step), arr, len(lrs), float(beta1), float(beta2), float(eps),
                                 _stream(step)), "anr_adam_step")


# ---- the explicit training step (fused_step.py): csrc/train_step.hip
def zero_fill(t: torch.Tensor) -> torch.Tensor:
    """t[:] = 0 by the library's own fill kernel (a memset node of a captured graph went stale on ROCm 7.2; a framework
    fill is a launch the step's graph should not hold)."""
    lib = _lib.load()
    if not t.is_cuda or not t.is_contiguous():
        raise RuntimeError("zero_fill: a contiguous tensor on the GPU")
    _lib.check(lib.anr_zero_fill(_ptr(t), t.numel() * t.element_size(), _stream(t)), "anr_zero_fill")
    return t


def add_inplace(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """dst += src (fp32, contiguous, same size) by a kernel of the library (no framework launch inside the training step)."""
    lib = _lib.load()
    if not (torch.is_tensor(dst) and dst.is_contiguous()) or dst.numel() != src.numel():
        raise ValueError("add_inplace: dst must be contiguous and as long as src")
    dst, src = _dev(dst, "dst"), _dev(src, "src")
    _lib.check(lib.anr_add_inplace(_ptr(dst), _ptr(src), dst.numel(), _stream(dst)), "anr_add_inplace")
    return dst


def add_segments(pairs) -> None:
    """dst += src (fp32, equal sizes) for every (dst, src) of `pairs` in ONE launch (anr_add_segments)."""
    import ctypes as C
    lib = _lib.load()
    pairs = [(d, s) for d, s in pairs if d.numel()]
    for d, s in pairs:
        if not (d.is_cuda and s.is_cuda and d.is_contiguous() and s.is_contiguous() and d.dtype == s.dtype == torch.float32
                and d.numel() == s.numel() and d.device == s.device):
            raise ValueError("add_segments: contiguous fp32 device tensors of equal size")
    for i in range(0, len(pairs), 24):
        part = pairs[i:i + 24]
        n = len(part)
        dst = (C.c_void_p * n)(*[d.data_ptr() for d, _ in part])
        src = (C.c_void_p * n)(*[s_.data_ptr() for _, s_ in part])
        nf = (C.c_int64 * n)(*[d.numel() for d, _ in part])
        _lib.check(lib.anr_add_segments(dst, src, nf, n, _stream(part[0][0])), "anr_add_segments")


def zero_segments(tensors) -> None:
    """t.zero_() for every tensor of `tensors` (contiguous, on the current device) in ONE launch (anr_zero_segments)."""
    import ctypes as C
    lib = _lib.load()
    ts = [t for t in tensors if t is not None and t.numel()]
    for i in range(0, len(ts), 24):
        part = ts[i:i + 24]
        for t in part:
            if not (t.is_cuda and t.is_contiguous() and t.device == part[0].device):
                raise ValueError("zero_segments: contiguous tensors on one device")
        n = len(part)
        dst = (C.c_void_p * n)(*[t.data_ptr() for t in part])
        nb = (C.c_int64 * n)(*[t.numel() * t.element_size() for t in part])
        _lib.check(lib.anr_zero_segments(dst, nb, n, _stream(part[0])), "anr_zero_segments")


def copy_segments(pairs) -> None:
    """dst.copy_(src) for every (dst, src) of `pairs` (contiguous tensors of equal byte size on the current device) in ONE launch
    of the library (anr_copy_segments): the batch of a training step moving into the buffers a captured step replays from."""
    import ctypes as C
    lib = _lib.load()
    pairs = [(d, s) for d, s in pairs if d.numel()]
    for i in range(0, len(pairs), 24):
        part = pairs[i:i + 24]
        for d, s in part:
            if not (d.is_cuda and s.is_cuda and d.is_contiguous() and s.is_contiguous() and d.device == s.device
                    and d.numel() * d.element_size() == s.numel() * s.element_size()):
                raise ValueError("copy_segments: contiguous device tensors of equal byte size")
        n = len(part)
        src = (C.c_void_p * n)(*[s.data_ptr() for _, s in part])
        dst = (C.c_void_p * n)(*[d.data_ptr() for d, _ in part])
        nb = (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in part])
        _lib.check(lib.anr_copy_segments(src, dst, nb, n, _stream(part[0][0])), "anr_copy_segments")


DRAW_STATE_WORDS = 35                                         # include/animnerf_hip.h: ANR_DRAW_STATE_WORDS


def train_draws(state: torch.Tensor, *, n_t=0, t_scale=1.0, n_nc=0, n_u=0, n_nf=0, noise_scale=1.0, verts_template=None,
                point_scale=0.0, neighbour_scale=0.0, quads: Optional[torch.Tensor] = None):
    """Every random number of one training step in one launch (anr_train_draws): -> dict(t_rand[n_t], noise_c[n_nc],
    u_fine[n_u], noise_f[n_nf], n0, n1 [like verts_template], pair[2 x verts_template rows, 3]); absent ones None.
    state: int64[DRAW_STATE_WORDS] on the device = (seed, step counter, tickets: zero); the kernel advances the counter."""
    lib = _lib.load()
    state = _dev(state, "state", torch.int64)
    if state.numel() < DRAW_STATE_WORDS:
        raise ValueError(f"train_draws: state holds {state.numel()} words, needs {DRAW_STATE_WORDS}")
    dev = state.device
    new = lambda n: torch.empty(n, dtype=torch.float32, device=dev) if n else None
    o = dict(t_rand=new(n_t), noise_c=new(n_nc), u_fine=new(n_u), noise_f=new(n_nf), n0=None, n1=None, pair=None)
    p = _lib.AnrDrawPlan()
    for k in ("t_rand", "noise_c", "u_fine", "noise_f"):
        setattr(p, k, None if o[k] is None else o[k].data_ptr())
    p.n_t, p.n_nc, p.n_u, p.n_nf = n_t, n_nc, n_u, n_nf
    p.t_scale, p.noise_scale = float(t_scale), float(noise_scale)
    vt = None
    if verts_template is not None:
        vt = _dev(verts_template, "verts_template")
        o["n0"], o["n1"] = torch.empty_like(vt), torch.empty_like(vt)
        o["pair"] = torch.empty(2 * vt.numel() // 3, 3, dtype=torch.float32, device=dev)
        p.verts_template, p.n_v3 = vt.data_ptr(), vt.numel()
        p.point_scale, p.neighbour_scale = float(point_scale), float(neighbour_scale)
        p.n0, p.n1, p.pair = o["n0"].data_ptr(), o["n1"].data_ptr(), o["pair"].data_ptr()
        if quads is not None:                                   # [>= 4 * pair rows, 4]: `pair` as tangent-mode quads (tangent_quads)
            quads = _dev(quads, "quads")
            if quads.numel() < 16 * o["pair"].shape[0]:
                raise ValueError("train_draws: quads holds fewer than four rows per point of the pair")
            p.quads = quads.data_ptr()
    if n_t + n_nc + n_u + n_nf + int(p.n_v3) == 0:               # perturb = 0 and no normals term: nothing random in the step
        return o
    _lib.check(lib.anr_train_draws(_ptr(state), C.byref(p), _stream(state)), "anr_train_draws")
    return o


def gather_frame_params(frame_idx, betas_w, go_w, bp_w, tr_w):
    """BodyModelParams.forward as one launch: -> betas[bs,10], pose[bs,72], transl[bs,3]."""
    lib = _lib.load()
    frame_idx = _dev(frame_idx, "frame_idx", torch.int64)
    betas_w, go_w, bp_w, tr_w = _dev(betas_w, "betas"), _dev(go_w, "global_orient"), _dev(bp_w, "body_pose"), _dev(tr_w, "transl")
    bs, dev = frame_idx.numel(), frame_idx.device
    betas, pose, transl = (torch.empty(bs, c, dtype=torch.float32, device=dev) for c in (10, 72, 3))
    _lib.check(lib.anr_gather_frame_params(_ptr(frame_idx), bs, _ptr(betas_w), betas_w.shape[0], _ptr(go_w), _ptr(bp_w), _ptr(tr_w),
                                           _ptr(betas), _ptr(pose), _ptr(transl), _stream(betas)), "anr_gather_frame_params")
    return betas, pose, transl


def scatter_frame_param_grads(frame_idx, grads, table_rows, betas_rows, d_betas, d_go, d_bp, d_tr):
    """grads[bs,85] -> the four tables' gradient buffers, written whole (None: skipped)."""
    lib = _lib.load()
    frame_idx, grads = _dev(frame_idx, "frame_idx", torch.int64), _dev(grads, "grads")
    for t in (d_betas, d_go, d_bp, d_tr):
        if t is not None and (not t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32):
            raise RuntimeError("scatter_frame_param_grads: contiguous float32 gradient buffers on the GPU")
    _lib.check(lib.anr_scatter_frame_param_grads(_ptr(frame_idx), _ptr(grads), frame_idx.numel(), int(table_rows), int(betas_rows),
                                                 _ptr(d_betas), _ptr(d_go), _ptr(d_bp), _ptr(d_tr), _stream(grads)),
               "anr_scatter_frame_param_grads")


def to_root_frame_from_chain(A, verts, joints, T):
    """`to_root_frame` with the root transforms read in place from the joint chain A[bs,J,4,4] (joints_transform[:, 0])."""
    lib = _lib.load()
    A, verts, joints, T = _dev(A, "joints_transform"), _dev(verts, "verts"), _dev(joints, "joints"), _dev(T, "T")
    bs, V, J = verts.shape[0], verts.shape[1], joints.shape[1]
    g_inv = torch.empty(bs, 4, 4, dtype=torch.float32, device=A.device)
    g_root = torch.empty_like(g_inv)
    v2, j2, T2 = torch.empty_like(verts), torch.empty_like(joints), torch.empty_like(T)
    _lib.check(lib.anr_to_root_frame_strided(_ptr(A), 16 * A.shape[1], _ptr(verts), _ptr(joints), _ptr(T), bs, V, J, _ptr(g_inv),
                                             _ptr(g_root), _ptr(v2), _ptr(j2), _ptr(T2), _stream(T2)), "anr_to_root_frame")
    return g_inv, g_root, v2, j2, T2


_COMPACT_STATE = {}             # (device, stream) -> the zeroed look-back state of anr_compact_ordered_single (it leaves it zero)


def compact_state(total: int, device) -> torch.Tensor:
    """Look-back state (int64 words, NOT zeroed) for one anr_compact_ordered_single call over `total` entries: the caller owns
    it and zeroes it in front of every call (the explicit training step: one more segment of its anr_zero_segments launch)."""
    words = int(_lib.load().anr_compact_state_words(int(total)))
    return torch.empty(max(words, 1), dtype=torch.int64, device=device)


def compact_ordered_riders(pts: torch.Tensor, fg: Optional[torch.Tensor] = None, bg: Optional[torch.Tensor] = None, single: bool = False,
                           state: Optional[torch.Tensor] = None):
    """`compact_ordered` with the prior points fg[rows,n_fg,3] / bg[rows,n_bg,3] appended as valid samples n .. n + n_r - 1, per
    frame its foreground then its background points (index / pos: n + n_r entries).
    single + state: the single-pass kernel's look-back state (`compact_state`), ZEROED BY THE CALLER in front of this call — the
    kernel needs it zero on entry and restores it only in its last block, so a state that outlives a faulted or aborted
    launch must not be trusted; without `state` a per-(device, stream) tensor kept here is used on that trust (eager calls)."""
    lib = _lib.load()
    pts = _dev(pts, "pts")
    n = pts.numel() // 4
    fg = None if fg is None else _dev(fg, "fg_points")
    bg = None if bg is None else _dev(bg, "bg_points")
    rows = 0 if (fg is None and bg is None) else (fg if fg is not None else bg).shape[0]
    n_fg, n_bg = (0 if fg is None else fg.shape[1]), (0 if bg is None else bg.shape[1])
    if fg is not None and bg is not None and fg.shape[0] != bg.shape[0]:
        raise ValueError("fg / bg prior points: one row per frame each")
    tot = n + rows * (n_fg + n_bg)
    index = torch.empty(tot, dtype=torch.int32, device=pts.device)
    pos = torch.empty(tot, dtype=torch.int32, device=pts.device)
    pts_c = torch.empty(-(-tot // 64) * 64, 4, dtype=torch.float32, device=pts.device)
    count = torch.empty(2, dtype=torch.int32, device=pts.device)
    if single:
        # one launch (chained scan); the look-back state lives per (device, stream): calls on one stream follow each other
        key = (pts.device.index, torch.cuda.current_stream(pts.device).cuda_stream)
        words = int(lib.anr_compact_state_words(tot))
        if state is not None:
            state = _dev(state, "state", torch.int64)
            if state.numel() < words:
                raise ValueError(f"compact_ordered_riders: state of {state.numel()} words, {words} needed")
        else:
            state = _COMPACT_STATE.get(key)
            if state is None or state.numel() < words:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("compact_ordered_riders(single=True): the look-back state must exist before a capture (run one eager step, or pass state=)")
                state = _COMPACT_STATE[key] = torch.zeros(max(words, 1024), dtype=torch.int64, device=pts.device)
        with _timed("compact_ordered", tot, tot * 24):
            _lib.check(lib.anr_compact_ordered_single(_ptr(pts), n, _ptr(fg), n_fg, _ptr(bg), n_bg, rows, _ptr(index), _ptr(pos), _ptr(pts_c),
                                                      _ptr(count), _ptr(state), _stream(pts)), "anr_compact_ordered_single")
        return index, pos, pts_c, count
    ws = torch.empty(lib.anr_compact_ws_ints(tot), dtype=torch.int32, device=pts.device)
    with _timed("compact_ordered", tot, tot * 24):
        _lib.check(lib.anr_compact_ordered_riders(_ptr(pts), n, _ptr(fg), n_fg, _ptr(bg), n_bg, rows, _ptr(index), _ptr(pos), _ptr(pts_c),
                                                  _ptr(count), _ptr(ws), _stream(pts)), "anr_compact_ordered")
    return index, pos, pts_c, count


def merge_backward2(g_a, g_b, perm_u8, Kc: int) -> torch.Tensor:
    """dL/dz_coarse[R,Kc] from (g_a + g_b)[R,K] and the byte permutation of sample_fine_merge(perm_u8=True)."""
    lib = _lib.load()
    g_a, perm_u8 = _dev(g_a, "g_a"), _dev(perm_u8, "perm", torch.uint8)
    g_b = None if g_b is None else _dev(g_b, "g_b")
    R, K = perm_u8.shape
    d = torch.empty(R, Kc, dtype=torch.float32, device=g_a.device)
    _lib.check(lib.anr_merge_backward2(_ptr(g_a), _ptr(g_b), _ptr(perm_u8), R, K, Kc, _ptr(d), _stream(d)), "anr_merge_backward2")
    return d


def sample_coarse_backward_acc(d_rays_acc, steps, t_rand, g_a, g_b=None, g_c=None, dfar_a=None, dfar_b=None) -> None:
    """d_rays_acc[R,8] columns 6, 7 += d(near', far') of sample_coarse from (g_a + g_b + g_c)[R,K], + dfar_a + dfar_b."""
    lib = _lib.load()
    d_rays_acc, steps, g_a = _dev(d_rays_acc, "d_rays_acc"), _dev(steps, "steps"), _dev(g_a, "g_a")
    opt = lambda t, nm: None if t is None else _dev(t, nm)
    t_rand, g_b, g_c, dfar_a, dfar_b = opt(t_rand, "t_rand"), opt(g_b, "g_b"), opt(g_c, "g_c"), opt(dfar_a, "dfar_a"), opt(dfar_b, "dfar_b")
    K = steps.numel()
    R = g_a.numel() // K
    _lib.check(lib.anr_sample_coarse_backward_acc(_ptr(g_a), _ptr(g_b), _ptr(g_c), _ptr(steps), _ptr(t_rand), _ptr(dfar_a), _ptr(dfar_b),
                                                  R, K, _ptr(d_rays_acc), _stream(d_rays_acc)), "anr_sample_coarse_backward_acc")


def warp_backward_acc(d_pts, rays, z, o2c, nbr_idx, nbr_w, d_o2c_acc, d_rays_acc, pos=None) -> torch.Tensor:
    """`warp_backward` adding into caller-owned accumulators d_o2c_acc[bs,V,4,4] / d_rays_acc[bs,R,8] (zeroed once per step:
    the coarse and the fine pass both add there); -> d_z[bs,R,K].
    pos: d_pts holds the rows of the compacted list of valid samples, pos[bs*R*K] maps a sample to its row or -1."""
    lib = _lib.load()
    if pos is not None:
        d_pts, rays, z, o2c = _dev(d_pts, "d_pts"), _dev(rays, "rays"), _dev(z, "z"), _dev(o2c, "ober2cano")
        nbr_idx, nbr_w, pos = _dev(nbr_idx, "nbr_idx", torch.int32), _dev(nbr_w, "nbr_w"), _dev(pos, "pos", torch.int32)
        d_o2c_acc, d_rays_acc = _dev(d_o2c_acc, "d_o2c_acc"), _dev(d_rays_acc, "d_rays_acc")
        bs, R, K = z.shape
        assert pos.numel() >= bs * R * K
        d_z = torch.empty_like(z)
        with _timed("warp_backward", bs * R * K):
            _lib.check(lib.anr_warp_backward_compact(_ptr(d_pts), _ptr(pos), _ptr(rays), rays.shape[-1], _ptr(z), K, _ptr(o2c), _ptr(nbr_idx),
                                                     _ptr(nbr_w), bs, o2c.shape[1], R * K, _ptr(d_o2c_acc), _ptr(d_rays_acc), _ptr(d_z),
                                                     _stream(z)), "anr_warp_backward_compact")
        return d_z
    d_pts, rays, z, o2c = _dev(d_pts, "d_pts"), _dev(rays, "rays"), _dev(z, "z"), _dev(o2c, "ober2cano")
    nbr_idx, nbr_w = _dev(nbr_idx, "nbr_idx", torch.int32), _dev(nbr_w, "nbr_w")
    d_o2c_acc, d_rays_acc = _dev(d_o2c_acc, "d_o2c_acc"), _dev(d_rays_acc, "d_rays_acc")
    bs, R, K = z.shape
    d_z = torch.empty_like(z)
    with _timed("warp_backward", bs * R * K):
        _lib.check(lib.anr_warp_backward(_ptr(d_pts), _ptr(rays), rays.shape[-1], _ptr(z), K, _ptr(o2c), _ptr(nbr_idx), _ptr(nbr_w), bs,
                                         o2c.shape[1], R * K, _ptr(d_o2c_acc), _ptr(d_rays_acc), _ptr(d_z), _stream(z)),
                   "anr_warp_backward")
    return d_z


def grid_points_cells(N: int, x_range, y_range, z_range, center: torch.Tensor, cells: torch.Tensor):
    """The grid points of the listed 8^3-voxel cells (int32 cell ids): -> pts[L*512,4], vox[L*512] int32 (flat grid indices)."""
    lib = _lib.load()
    center, cells = _dev(center.reshape(-1), "center"), _dev(cells, "cells", torch.int32)
    L = cells.numel()
    pts = torch.empty(L * 512, 4, dtype=torch.float32, device=center.device)
    vox = torch.empty(L * 512, dtype=torch.int32, device=center.device)
    _lib.check(lib.anr_grid_points_cells(N, float(x_range[0]), float(x_range[1]), float(y_range[0]), float(y_range[1]), float(z_range[0]),
                                         float(z_range[1]), _ptr(center), _ptr(cells), L, _ptr(pts), _ptr(vox), _stream(pts)),
               "anr_grid_points_cells")
    return pts, vox


def scatter_relu(values: torch.Tensor, vox: torch.Tensor, out: torch.Tensor, first: int) -> None:
    """out[vox[t] - first] = relu(values[t]) for the voxels that fall inside out."""
    lib = _lib.load()
    values, vox, out = _dev(values, "values"), _dev(vox, "vox", torch.int32), _dev(out, "out")
    with _timed("scatter_relu", values.numel(), values.numel() * 12):
        _lib.check(lib.anr_scatter_relu(_ptr(values), _ptr(vox), values.numel(), int(first), out.numel(), _ptr(out), _stream(out)),
                   "anr_scatter_relu")


def knn_within(verts: torch.Tensor, xyz: torch.Tensor, radius: float, index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d1[bs,N]: distance to the nearest vertex where below `radius`, +inf elsewhere (the exact search started from that bound)."""
    lib = _lib.load()
    verts, xyz = _dev(verts, "verts"), _dev(xyz, "xyz")
    bs, V, _ = verts.shape
    N = xyz.shape[1]
    if index is None:
        index = knn_index_build(verts, morton_order(verts[0]).to(verts.device))
    index = _dev(index, "knn_index", torch.uint8)
    d1 = torch.empty(bs, N, dtype=torch.float32, device=xyz.device)
    with _timed("knn_within", bs * N):
        _lib.check(lib.anr_knn_within(_ptr(index), _ptr(xyz), bs, V, N, float(radius), _ptr(d1), _stream(xyz)), "anr_knn_within")
    return d1
