"""On-disk formats and host-side sampling either side of the rendering path.

Restates, without cv2 / torchvision / Lightning:
  * checkpoints: `extract_model_state_dict`, `load_ckpt`, `load_hparams` (utils/__init__.py:78-105);
  * People-Snapshot style folders (tools/people_snapshot.py:56-91, tools/prepare_template.py:91-105):
    `cam{ID:03d}/camera.pkl`, `smpls/{frame:06d}.pkl`, `smpl_template.pkl` and what AnimNeRFDatasets makes of them
    (datasets/anim_nerf_dataset.py:124-233): camera rescale, camera -> c2w with the diag(1,-1,-1) flip, rays,
    per-frame body parameters, template parameters, foreground / background prior points;
  * training pixel sampling `get_pixelcoords` (datasets/anim_nerf_dataset.py:10-54);
  * the novel-view orbit (novel_view.py:192-198).
Image decoding / undistortion (cv2.imread, cv2.undistort) is not restated: callers hand over arrays.
"""
from __future__ import annotations

import argparse
import math
import os
import pickle
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from .rays import gen_rays


# ----------------------------------------------------------------------------------------------- pickles
def load_pickle_file(pkl_path):
    """utils/util.py:33-36 (latin1: the SMPL-era pickles were written by Python 2)."""
    with open(pkl_path, "rb") as f:
        return pickle.load(f, encoding="latin1")


def write_pickle_file(pkl_path, data_dict):
    """utils/util.py:39-41."""
    with open(pkl_path, "wb") as fp:
        pickle.dump(data_dict, fp, protocol=2)


# ----------------------------------------------------------------------------------------------- checkpoints
def extract_model_state_dict(ckpt, model_name="model", prefixes_to_ignore=()):
    """utils/__init__.py:78-93.  `ckpt` is a path or an already loaded dict; a Lightning checkpoint keeps the weights
    under 'state_dict' with the attribute name of the module as prefix (`anim_nerf.` for AnimNeRFSystem)."""
    checkpoint = torch.load(ckpt, map_location="cpu", weights_only=False) if isinstance(ckpt, (str, os.PathLike)) else ckpt
    if "state_dict" in checkpoint:
        checkpoint = checkpoint["state_dict"]
    out = {}
    for k, v in checkpoint.items():
        if not k.startswith(model_name + "."):
            continue
        k = k[len(model_name) + 1:]
        if any(k.startswith(p) for p in prefixes_to_ignore):
            continue
        out[k] = v
    return out


def load_ckpt(model, ckpt, model_name="model", prefixes_to_ignore=()):
    """utils/__init__.py:95-99: checkpoint entries override the module's own state, the rest stays."""
    state = model.state_dict()
    state.update(extract_model_state_dict(ckpt, model_name, prefixes_to_ignore))
    model.load_state_dict(state)


def load_hparams(ckpt):
    """utils/__init__.py:101-105."""
    checkpoint = torch.load(ckpt, map_location="cpu", weights_only=False) if isinstance(ckpt, (str, os.PathLike)) else ckpt
    return argparse.Namespace(**checkpoint["hyper_parameters"])


# ----------------------------------------------------------------------------------------------- cameras
def rescale_camera(cam: dict, img_wh: Tuple[int, int]) -> dict:
    """datasets/anim_nerf_dataset.py:176-179 (and novel_view.py:55-58): intrinsics follow the resize to img_wh.
    Returns a new dict; the reference mutates its argument."""
    cam = dict(cam)
    scale = np.array([img_wh[0] / cam["width"], img_wh[1] / cam["height"]])
    cam["camera_f"] = np.asarray(cam["camera_f"]) * scale
    cam["camera_c"] = np.asarray(cam["camera_c"]) * scale
    cam["height"], cam["width"] = img_wh[1], img_wh[0]
    return cam


def camera_to_c2w(cam: dict) -> torch.Tensor:
    """datasets/anim_nerf_dataset.py:207-224: world->camera (R, t) in the OpenCV convention -> camera->world [3,4] in the
    OpenGL convention the ray generator expects (flip y and z)."""
    flip = np.diag([1.0, -1.0, -1.0])                       # OpenCV (x right, y down, z forward) -> OpenGL axes
    rot = (flip @ np.asarray(cam["R"], dtype=np.float64)).T    # camera -> world rotation
    centre = rot @ -(flip @ np.asarray(cam["t"], dtype=np.float64))
    c2w = np.concatenate([rot, centre[:, None]], axis=1).astype(np.float32)
    return torch.from_numpy(c2w)


def camera_rays(cam: dict, near: float = 0.1, far: float = 10.0, device=None) -> torch.Tensor:
    """AnimNeRFDatasets.get_rays (datasets/anim_nerf_dataset.py:207-226) -> rays[H, W, 8] (on `device`: HIP kernel)."""
    c2w = camera_to_c2w(cam)
    if device is not None:
        c2w = c2w.to(device)
    return gen_rays(c2w, int(cam["height"]), int(cam["width"]), list(np.asarray(cam["camera_f"], dtype=np.float64)),
                    near, far, list(np.asarray(cam["camera_c"], dtype=np.float64)))


def load_camera(root_dir: str, cam_id: int = 0) -> dict:
    """datasets/anim_nerf_dataset.py:162-165: {R[3,3], t[3], camera_f[2], camera_c[2], camera_k[5], height, width}."""
    return load_pickle_file(os.path.join(root_dir, "cam{:0>3d}".format(cam_id), "camera.pkl"))


def orbit_transforms(n_views: int = 120, angle: float = 0.0) -> torch.Tensor:
    """novel_view.py:192-198: P_i = R_y(2 pi i / n_views) R_x(-angle deg) as [n_views, 4, 4]; applied to the rays in
    the body frame (`batched_inference(..., P=P_i[None, None])`)."""
    a = -math.radians(angle)
    Rz = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])   # cv2.Rodrigues([a,0,0])
    out = np.tile(np.eye(4, dtype=np.float32), (n_views, 1, 1))
    for i in range(n_views):
        b = 2 * np.pi * i / n_views
        Ry = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
        out[i, :3, :3] = Ry @ Rz
    return torch.from_numpy(out)


# ----------------------------------------------------------------------------------------------- body parameters
_SMPL_KEYS = ("betas", "global_orient", "body_pose", "transl")


def load_body_model_params(root_dir: str, frame_id: int, model_type: str = "smpl") -> Dict[str, torch.Tensor]:
    """datasets/anim_nerf_dataset.py:134-160 for model_type 'smpl': smpls/{frame:06d}.pkl -> betas[10],
    global_orient[3], body_pose[69], transl[3] (fp32, unbatched)."""
    if model_type != "smpl":
        raise ValueError(f"Unknown model type {model_type}, exiting!")
    params = load_pickle_file(os.path.join(root_dir, f"{model_type}s", "{:0>6}.pkl".format(frame_id)))
    return {k: torch.from_numpy(np.asarray(params[k])).float() for k in _SMPL_KEYS}


def load_template(root_dir: str, model_type: str = "smpl"):
    """datasets/anim_nerf_dataset.py:122-131: {model_type}_template.pkl -> (params with the `_template` suffix the
    training batch uses, fg_points = points with signed distance < -0.02, bg_points = points with distance > 0.10)."""
    t = load_pickle_file(os.path.join(root_dir, f"{model_type}_template.pkl"))
    params = {k + "_template": torch.from_numpy(np.asarray(t[k])).float() for k in _SMPL_KEYS}
    pts, dist = np.asarray(t["points"]), np.asarray(t["distances"])
    return params, torch.from_numpy(pts[dist < -0.02]).float(), torch.from_numpy(pts[dist > 0.10]).float()


def sample_prior_points(fg_points: torch.Tensor, bg_points: torch.Tensor, num_points: int = 128, generator=None):
    """AnimNeRFDatasets.get_points (datasets/anim_nerf_dataset.py:228-233): random picks + N(0, 0.01) jitter."""
    fg = fg_points[torch.randint(0, fg_points.shape[0], (num_points,), generator=generator)]
    fg = fg + torch.randn(fg.shape, generator=generator) * 0.01
    bg = bg_points[torch.randint(0, bg_points.shape[0], (num_points,), generator=generator)]
    bg = bg + torch.randn(bg.shape, generator=generator) * 0.01
    return fg, bg


def frame_index(frame_ids: Sequence[int]) -> Dict[int, int]:
    """datasets/anim_nerf_dataset.py:111-115: frame id -> row of the BodyModelParams / latent-code tables."""
    return {f: i for i, f in enumerate(frame_ids)}


# ----------------------------------------------------------------------------------------------- pixel sampling
def _rank_filter(mask: np.ndarray, k: int, take_max: bool) -> np.ndarray:
    """cv2.erode / cv2.dilate with a k x k box, default anchor (k // 2) and default border (the border never wins:
    +inf for erode, -inf for dilate), separable running min / max."""
    out = np.asarray(mask, dtype=np.float64)
    pad_val = -np.inf if take_max else np.inf
    a = k // 2                                   # window = [x - a, x - a + k - 1]
    for axis in (0, 1):
        n = out.shape[axis]
        pad = [(0, 0), (0, 0)]
        pad[axis] = (a, k - 1 - a)
        p = np.pad(out, pad, constant_values=pad_val)
        win = np.lib.stride_tricks.sliding_window_view(p, k, axis=axis)
        out = win.max(-1) if take_max else win.min(-1)
        assert out.shape[axis] == n
    return out.astype(np.asarray(mask).dtype)


def get_pixelcoords(H, W, mask=None, subsampletype="foreground_pixel", subsamplesize=32, fore_rate=0.9, fore_erode=3):
    """datasets/anim_nerf_dataset.py:10-54 -> pixelcoords[n, 2] = (row, col).  Draws from numpy's global generator in the
    reference's order (one `choice` per pixel pool), so `np.random.seed(s)` reproduces the reference's picks.
    'foreground_pixel' (every shipped yaml): fore_rate of the patch from the eroded mask, the rest from the band between
    the mask dilated by fore_erode and by 64.  The morphology is restated from cv2's documented semantics (cv2 is not
    available where this was written: that mode is checked against a brute-force min / max filter, not against cv2)."""
    n_patch = subsamplesize * subsamplesize

    def draw(rows, cols, count):                       # with replacement, as the reference does
        pick = np.random.choice(rows.shape[0], count, replace=True)
        return rows[pick], cols[pick]

    if subsampletype == "pixel":
        rr, cc = np.meshgrid(np.arange(0, H), np.arange(0, W), indexing="ij")
        rows, cols = draw(rr.ravel(), cc.ravel(), n_patch)
    elif subsampletype == "foreground_pixel":
        m = np.asarray(mask)
        m = m[..., 0] if m.ndim == 3 else m
        core = _rank_filter(m, fore_erode, take_max=False)
        band = _rank_filter(m, 64, take_max=True) - _rank_filter(m, fore_erode, take_max=True)
        n_fore = int(n_patch * fore_rate)
        fr, fc = draw(*np.where(core > 0), n_fore)
        br, bc = draw(*np.where(band > 0), n_patch - n_fore)
        rows, cols = np.concatenate((fr, br)), np.concatenate((fc, bc))
    else:                                              # every pixel, row-major
        rows, cols = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    return np.stack((rows.reshape(-1), cols.reshape(-1)), axis=-1)


def subsample_training_pixels(rays, rgbs, alphas, coords, subsamplesize=32):
    """datasets/anim_nerf_dataset.py:262-266: gather the sampled pixels into [s, s, C] patches."""
    r, c = coords[:, 0], coords[:, 1]
    s = subsamplesize
    return rays[r, c].view(s, s, 8), rgbs[r, c].view(s, s, 3), alphas[r, c].view(s, s, 1)


def load_mixamo_smpl(actions_dir, action_type="0007", skip=1):
    """novel_pose.py:26-41: a Mixamo action retargeted to SMPL, `<actions_dir>/<action_type>/result.pkl` = {anim_len,
    smpl_array[anim_len * 72], cam_array[anim_len, 3]} -> per frame {global_orient[3], body_pose[69], transl[3]}; the root
    translation of a frame is (cam[1], cam[2], 0), as the reference takes it."""
    import os
    import pickle
    with open(os.path.join(actions_dir, action_type, "result.pkl"), "rb") as f:
        result = pickle.load(f, encoding="latin1")
    return mocap_frames(result, skip)


def mocap_frames(result, skip=1):
    """the frame list of novel_pose.py:29-41 from the already loaded dict"""
    import numpy as np
    anim_len = int(result["anim_len"])
    pose = np.asarray(result["smpl_array"], dtype=np.float32).reshape(anim_len, -1)
    cam = np.asarray(result["cam_array"], dtype=np.float32)
    return [{"cam": cam[i], "global_orient": pose[i, :3], "body_pose": pose[i, 3:72],
             "transl": np.array([cam[i, 1], cam[i, 2], 0.0], dtype=np.float32)} for i in range(0, anim_len, skip)]
