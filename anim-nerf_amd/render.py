"""Frame drivers: the ray-chunk loop the reference repeats in train.py:189-215,
novel_view.py:78-98, novel_pose.py:43-80 and extract_mesh.py:49-61, plus the one-process-per-GPU
ray sharding used by bench.py (rays are independent, no collective on the data path).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, Optional, Tuple

import torch

from .anim_nerf import batch_transform


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n units owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@torch.no_grad()
def batched_inference(volume_renderer, anim_nerf, rays, body_model_params, body_model_params_template,
                      latent_code=None, P=None, chunk=2048) -> Dict[str, torch.Tensor]:
    """novel_view.py:78-98.  rays[bs,n_rays,8] -> dict of [bs,n_rays,C]."""
    anim_nerf.set_body_model(body_model_params, body_model_params_template)
    rays = anim_nerf.convert_to_body_model_space(rays)
    anim_nerf.clac_ober2cano_transform()
    if latent_code is not None:
        anim_nerf.set_latent_code(latent_code)
    if P is not None:
        rays = rays.clone()
        rays[:, :, 0:3] = batch_transform(P, rays[:, :, 0:3], pad_ones=True)
        rays[:, :, 3:6] = batch_transform(P, rays[:, :, 3:6], pad_ones=False)
    return render_prepared(volume_renderer, anim_nerf, rays, chunk=chunk, perturb=0.0)


def render_prepared(volume_renderer, anim_nerf, rays, chunk=2048, perturb=0.0):
    """The hot loop alone: per-frame state already set on `anim_nerf`, rays already in the body frame."""
    n_rays = rays.shape[1]
    pieces = defaultdict(list)
    for i in range(0, n_rays, chunk):
        part = volume_renderer(anim_nerf, rays[:, i:i + chunk], perturb=perturb)
        for k, v in part.items():
            pieces[k].append(v)
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 1)) for k, v in pieces.items()}


def system_forward(volume_renderer, anim_nerf, rays, body_model_params, body_model_params_template,
                   latent_code=None, perturb=1.0, chunk=2048):
    """AnimNeRFSystem.forward (train.py:189-215): rays[bs,h,w,8] -> dict of [bs,h,w,C]."""
    bs, h, w = rays.shape[:3]
    flat = rays.view(bs, h * w, -1)
    if hasattr(anim_nerf, "frame_setup"):                      # (the three calls below as two launches on the GPU)
        flat = anim_nerf.frame_setup(body_model_params, body_model_params_template, flat)
    else:
        anim_nerf.set_body_model(body_model_params, body_model_params_template)
        flat = anim_nerf.convert_to_body_model_space(flat)
        anim_nerf.clac_ober2cano_transform()
    if latent_code is not None:
        anim_nerf.set_latent_code(latent_code)
    out = render_prepared(volume_renderer, anim_nerf, flat, chunk=chunk, perturb=perturb)
    return {k: v.view(bs, h, w, -1) for k, v in out.items()}


_CELL_CENTRES = {}


@torch.no_grad()
def sigma_grid(anim_nerf, N_grid=256, x_range=(-1.2, 1.2), y_range=(-1.2, 1.2), z_range=(-1.2, 1.2),
               chunk=1 << 22, rank=0, world=1, cells: Optional[bool] = None):
    """extract_mesh.py:152-158 on the fast path: relu(sigma) of the fine (if any) field on this rank's contiguous
    slab of the N^3 grid around the posed body's bounding-box centre.  Grid points are generated on the device,
    provably-empty voxels skip the neighbour search, the MLP stops at the sigma row.  Returns (sigma[count], first).
    Per-frame state (set_body_model / convert_to_body_model_space / clac_ober2cano_transform) must be set.

    cells (default: whenever N % 8 == 0 and the warp is on): empty space is skipped by 8^3-voxel CELLS before a single voxel is
    generated.  A voxel is valid only if its blended neighbour distance — never below its distance d1 to the nearest vertex —
    is under dis_threshold (models/anim_nerf.py:183), so a cell whose centre is at least dis_threshold + r from every vertex
    (r = the cell's half diagonal, ONE exact search per cell centre) holds only sigma = -1e5 -> relu = 0: its 512 voxels are
    never generated, warped or compacted.  The other cells' voxels go through the same kernels as before: same bits."""
    from . import ops
    total = N_grid ** 3
    lo, hi = shard_range(total, rank, world)
    center = (anim_nerf.verts.max(dim=1)[0] + anim_nerf.verts.min(dim=1)[0]) / 2.     # [1,3]
    net = anim_nerf._net(anim_nerf.use_fine)
    if cells is None:
        cells = bool(anim_nerf.use_unpose and anim_nerf.k_neigh == 4 and N_grid % 8 == 0 and N_grid <= 1288 and anim_nerf.evaluate_valid_only)
    if cells:
        dev = center.device
        C = N_grid // 8
        key = (N_grid, tuple(x_range), tuple(y_range), tuple(z_range), str(dev))
        base = _CELL_CENTRES.get(key)
        if base is None:                                             # constants of the grid: built once, not per call
            axis = lambda r: r[0] + (r[1] - r[0]) * (8.0 * torch.arange(C, device=dev, dtype=torch.float64) + 3.5) / (N_grid - 1)
            cx, cy, cz = axis(x_range), axis(y_range), axis(z_range)
            # cell (cj, ci, ck) -> centre (x[ci], y[cj], z[ck]) (np.meshgrid 'xy': array axis 0 runs over y)
            base = torch.stack(torch.broadcast_tensors(cx[None, :, None], cy[:, None, None], cz[None, None, :]), -1).reshape(1, -1, 3).float()
            if len(_CELL_CENTRES) > 8:
                _CELL_CENTRES.clear()
            _CELL_CENTRES[key] = base
        cen = base + center
        step = torch.tensor([(r[1] - r[0]) / (N_grid - 1) for r in (x_range, y_range, z_range)], dtype=torch.float64)
        radius = float((3.5 * step).norm()) + 1e-4                  # half diagonal of the cell's voxel positions (+ rounding)
        # (the search starts from the bound: a centre far from the body is settled by the index's 14 top boxes)
        bound = anim_nerf.dis_threshold + radius
        live = ops.knn_within(anim_nerf.verts[:1], cen, bound, index=anim_nerf.knn_index()[:1])[0] < bound
        if world > 1:                                                # this rank's slab: cells whose j range meets [lo, hi)
            cj = torch.arange(C ** 3, device=dev) // (C * C)
            live &= (cj >= lo // (8 * N_grid * N_grid)) & (cj <= (hi - 1) // (8 * N_grid * N_grid))
        ids = torch.nonzero(live)[:, 0].to(torch.int32)
        out = ops.zero_fill(torch.empty(hi - lo, dtype=torch.float32, device=dev))
        per = max(1, chunk // 512)
        for s in range(0, ids.numel(), per):
            pts, vox = ops.grid_points_cells(N_grid, x_range, y_range, z_range, center[0], ids[s:s + per])
            pts = anim_nerf.warped_points(xyz=pts.view(1, -1, 4), skip_far=True)
            ops.scatter_relu(net.eval_points(pts, sigma_only=True, only_valid=True), vox, out, lo)
        return out, lo
    out = torch.empty(hi - lo, dtype=torch.float32, device=center.device)
    for s in range(lo, hi, chunk):
        n = min(chunk, hi - s)
        pts = ops.grid_points(N_grid, x_range, y_range, z_range, center[0], s, n)
        if anim_nerf.use_unpose:
            pts = anim_nerf.warped_points(xyz=pts.view(1, n, 4), skip_far=True)
        # voxels farther than dis_threshold from the body are sigma = -1e5 -> relu = 0: the MLP runs on the others only
        out[s - lo:s - lo + n] = torch.relu_(net.eval_points(pts, sigma_only=True,
                                                             only_valid=anim_nerf.evaluate_valid_only))
    return out, lo


@torch.no_grad()
def sigma_grid_inference(anim_nerf, points, chunk=32 * 32 * 64):
    """extract_mesh.py:49-61: relu(sigma) of the (fine) field on explicit points[bs,nv,3]."""
    out = []
    for i in range(0, points.shape[1], chunk):
        _, s = anim_nerf(points[:, i:i + chunk], None, use_fine=anim_nerf.use_fine)
        out.append(torch.relu(s))
    return torch.cat(out, 1)


# ---------------------------------------------------------------------------------------------
# one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests)

def max_over_ranks(seconds: float, device=None) -> float:
    """Slowest rank's time: what a whole-job throughput is divided by.  No-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def gather_ray_shards(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Assemble per-rank results for `shard_range(n_total, rank, world)` slices (dim 1) on every rank.
    Image assembly only — never on the timed data path."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    width = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world))
    pad = torch.zeros(local.shape[0], width, *local.shape[2:], dtype=local.dtype, device=local.device)
    pad[:, :local.shape[1]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:, :shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0]]
                      for r, p in enumerate(parts)], 1)
