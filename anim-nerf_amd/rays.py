"""Ray generation with the reference's call signatures, on the HIP kernel `anr_ray_gen`.

`gen_rays` is the live generator (datasets/anim_nerf_dataset.py:72-85); `get_ray_directions` /
`get_rays` are the dead twins BASELINE.json names (utils/ray_utils.py:74-121: scalar focal,
principal point at (W/2, H/2)).
"""
from __future__ import annotations

import torch

from . import ops


def _f32(x, device):
    return torch.as_tensor(x, dtype=torch.float32, device=device).contiguous()


def gen_rays(c2w, H, W, focal, near, far, c=None):
    """-> rays[H,W,8] = [o(3), d(3), near, far] on c2w's device (must be a GPU)."""
    dev = c2w.device
    return ops.ray_gen(_f32(c2w, dev), H, W, _f32(focal, dev), near, far, None if c is None else _f32(c, dev))


def gen_ray_directions(H, W, focal, c=None, device="cuda"):
    """Camera-frame unit directions [H,W,3] (identity c2w)."""
    eye = torch.eye(3, 4, dtype=torch.float32, device=device)
    return gen_rays(eye, H, W, focal, 0.0, 0.0, c)[..., 3:6]


def get_ray_directions(H, W, focal, device="cuda"):
    return gen_ray_directions(H, W, [focal, focal], [W / 2, H / 2], device=device)


def get_rays(directions, c2w):
    """utils/ray_utils.py:99-121: rotate precomputed camera directions into the world."""
    rays_d = directions @ c2w[:, :3].T
    return c2w[:, 3].expand(rays_d.shape), rays_d
