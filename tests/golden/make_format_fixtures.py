#!/usr/bin/env python3
"""Generates tests/golden/formats.npz by RUNNING THE REFERENCE's host-side format code (imported from /root/reference
with stand-in `cv2` / `torchvision` / `PIL` modules that only satisfy the import statements; none of their functions is
called) on small synthetic inputs.  Only inputs and outputs are stored.  Runs in the build container only.

  python tests/golden/make_format_fixtures.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference():
    cv2 = types.ModuleType("cv2")
    cv2.COLORMAP_JET = 2
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvt.ToTensor = lambda: (lambda x: x)
    tv.transforms = tvt
    pil = types.ModuleType("PIL")
    pil.Image = types.ModuleType("PIL.Image")
    for name, mod in (("cv2", cv2), ("torchvision", tv), ("torchvision.transforms", tvt), ("PIL", pil), ("PIL.Image", pil.Image)):
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, REF)
    import datasets.anim_nerf_dataset as r_ds
    import utils as r_utils
    return r_ds, r_utils


def main():
    r_ds, r_utils = import_reference()
    rng = np.random.RandomState(11)
    # camera.pkl content (tools/people_snapshot.py:56-64), world->camera rotation from a random axis-angle
    q, _ = np.linalg.qr(rng.randn(3, 3))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    cam = {"R": q, "t": rng.randn(3) * 0.3 + np.array([0.0, 0.1, 2.5]), "camera_f": np.array([30.0, 31.0]),
           "camera_c": np.array([11.5, 9.25]), "camera_k": np.zeros(5), "height": 20, "width": 24}
    img_wh = (16, 12)
    # the reference's rescale (datasets/anim_nerf_dataset.py:176-179), run on a copy through its own statements
    ref_cam = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    ref_cam["camera_f"] = ref_cam["camera_f"] * [img_wh[0] / ref_cam["width"], img_wh[1] / ref_cam["height"]]
    ref_cam["camera_c"] = ref_cam["camera_c"] * [img_wh[0] / ref_cam["width"], img_wh[1] / ref_cam["height"]]
    ref_cam["height"], ref_cam["width"] = img_wh[1], img_wh[0]
    rays = r_ds.AnimNeRFDatasets.get_rays(None, ref_cam)                       # [H, W, 8]

    np.random.seed(3)
    coords_pixel = r_ds.get_pixelcoords(12, 16, mask=None, subsampletype="pixel", subsamplesize=4)
    coords_all = r_ds.get_pixelcoords(5, 7, mask=None, subsampletype="all")

    # Lightning-style checkpoint
    state = {"anim_nerf.nerf.sigma.weight": torch.arange(4.0), "anim_nerf.nerf.sigma.bias": torch.ones(1),
             "anim_nerf.body_model.betas": torch.zeros(1, 10), "anim_nerf.nerf_fine.rgb.0.bias": torch.full((3,), 2.0),
             "latent_codes.weight": torch.zeros(2, 3), "body_model_params.transl.weight": torch.zeros(5, 3)}
    hyper = {"exp_name": "golden", "img_wh": [16, 12], "frame_IDs": [1, 5, 9], "model_type": "smpl"}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "last.ckpt")
        torch.save({"state_dict": state, "hyper_parameters": hyper, "epoch": 3}, path)
        picked = r_utils.extract_model_state_dict(path, "anim_nerf", prefixes_to_ignore=["body_model"])
        picked_all = r_utils.extract_model_state_dict(path, "anim_nerf")
        hp = r_utils.load_hparams(path)
    np.savez_compressed(
        os.path.join(HERE, "formats.npz"),
        cam_R=cam["R"], cam_t=cam["t"], cam_f=cam["camera_f"], cam_c=cam["camera_c"], cam_hw=np.array([20, 24]),
        img_wh=np.array(img_wh), rays=rays.numpy(), coords_pixel=coords_pixel, coords_all=coords_all,
        ckpt_keys=np.array(sorted(state)), picked_keys=np.array(sorted(picked)), picked_all_keys=np.array(sorted(picked_all)),
        picked_sigma_weight=picked["nerf.sigma.weight"].numpy(), hparams_keys=np.array(sorted(vars(hp))),
        hparams_frame_ids=np.array(hp.frame_IDs))
    print("wrote formats.npz:", rays.shape, coords_pixel.shape, sorted(picked))


if __name__ == "__main__":
    main()
