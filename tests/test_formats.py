"""Host-side formats and sampling (anim_nerf_amd.data) against outputs of the reference's own code
(tests/golden/formats.npz, made by tests/golden/make_format_fixtures.py) and against independent restatements."""
import os
import tempfile

import numpy as np
import pytest
import torch

from helpers import golden


def _cam(g):
    return {"R": g["cam_R"], "t": g["cam_t"], "camera_f": g["cam_f"], "camera_c": g["cam_c"], "camera_k": np.zeros(5),
            "height": int(g["cam_hw"][0]), "width": int(g["cam_hw"][1])}


def test_camera_rescale_and_c2w_match_reference():
    from anim_nerf_amd import data
    from oracle import animnerf_oracle as orc
    g = golden("formats")
    cam = _cam(g)
    small = data.rescale_camera(cam, tuple(int(v) for v in g["img_wh"]))
    assert cam["width"] == 24 and small["width"] == 16 and small["height"] == 12          # the input is not mutated
    c2w = data.camera_to_c2w(small)
    rays = orc.make_rays(c2w, small["height"], small["width"], list(small["camera_f"]), 0.1, 10.0, list(small["camera_c"]))
    ref = torch.from_numpy(g["rays"])
    assert rays.shape == ref.shape
    torch.testing.assert_close(rays, ref, rtol=1e-6, atol=1e-6)
    # camera origin = -R^T t in the flipped frame
    torch.testing.assert_close(ref[0, 0, :3].double(), torch.from_numpy(-(cam["R"].T @ cam["t"])), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_camera_rays_on_device_match_reference():
    from anim_nerf_amd import data
    g = golden("formats")
    small = data.rescale_camera(_cam(g), tuple(int(v) for v in g["img_wh"]))
    rays = data.camera_rays(small, device=torch.device("cuda:0"))
    torch.testing.assert_close(rays.cpu(), torch.from_numpy(g["rays"]), rtol=1e-6, atol=1e-6)


def test_pixel_sampling_matches_reference():
    from anim_nerf_amd import data
    g = golden("formats")
    np.random.seed(3)
    assert np.array_equal(data.get_pixelcoords(12, 16, subsampletype="pixel", subsamplesize=4), g["coords_pixel"])
    assert np.array_equal(data.get_pixelcoords(5, 7, subsampletype="all"), g["coords_all"])


def test_foreground_sampling_morphology():
    """cv2.erode / cv2.dilate semantics restated: box min / max, anchor k // 2, border never wins."""
    from anim_nerf_amd import data
    rng = np.random.RandomState(0)
    m = (rng.rand(40, 50) > 0.6).astype(np.float32)
    m[10:30, 15:35] = 1.0
    for k in (3, 5, 64):
        for take_max in (False, True):
            got = data._rank_filter(m, k, take_max)
            a = k // 2
            want = np.empty_like(m)
            for i in range(m.shape[0]):
                for j in range(m.shape[1]):
                    win = m[max(i - a, 0):min(i - a + k, m.shape[0]), max(j - a, 0):min(j - a + k, m.shape[1])]
                    want[i, j] = win.max() if take_max else win.min()
            assert np.array_equal(got, want), (k, take_max)
    mask = np.zeros((96, 96, 1), np.float32)
    mask[30:70, 35:60] = 1.0
    np.random.seed(1)
    c = data.get_pixelcoords(96, 96, mask, "foreground_pixel", subsamplesize=8, fore_rate=0.9, fore_erode=3)
    assert c.shape == (64, 2)
    fore, back = c[:57], c[57:]                                   # int(64 * 0.9) = 57 foreground picks first
    assert (mask[fore[:, 0], fore[:, 1], 0] == 1).all()
    inside_eroded = (fore[:, 0] >= 31) & (fore[:, 0] <= 68) & (fore[:, 1] >= 36) & (fore[:, 1] <= 58)
    assert inside_eroded.all()
    assert (mask[back[:, 0], back[:, 1], 0] == 0).all()           # the band outside the dilated mask


def test_checkpoint_extraction_matches_reference():
    from anim_nerf_amd import data
    g = golden("formats")
    state = {k: torch.zeros(1) for k in g["ckpt_keys"].tolist()}
    state["anim_nerf.nerf.sigma.weight"] = torch.arange(4.0)
    ckpt = {"state_dict": state, "hyper_parameters": {"exp_name": "golden", "img_wh": [16, 12], "frame_IDs": [1, 5, 9],
                                                      "model_type": "smpl"}}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "last.ckpt")
        torch.save(ckpt, path)
        picked = data.extract_model_state_dict(path, "anim_nerf", prefixes_to_ignore=["body_model"])
        hp = data.load_hparams(path)
    assert sorted(picked) == g["picked_keys"].tolist()
    assert sorted(data.extract_model_state_dict(ckpt, "anim_nerf")) == g["picked_all_keys"].tolist()
    assert np.array_equal(picked["nerf.sigma.weight"].numpy(), g["picked_sigma_weight"])
    assert sorted(vars(hp)) == g["hparams_keys"].tolist() and hp.frame_IDs == g["hparams_frame_ids"].tolist()


def test_load_ckpt_into_module_and_folder_formats(smpl_table):
    """A reference-layout checkpoint loads into AnimNeRF (state-dict keys are the reference's); smpls/*.pkl,
    smpl_template.pkl and cam000/camera.pkl round-trip through the loaders."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import data
    torch.manual_seed(1)
    src = ana.AnimNeRF(body_model_table=smpl_table, freqs_dir=0, use_view=False, use_unpose=True, use_fine=True)
    dst = ana.AnimNeRF(body_model_table=smpl_table, freqs_dir=0, use_view=False, use_unpose=True, use_fine=True)
    ckpt = {"state_dict": {"anim_nerf." + k: v.clone() for k, v in src.state_dict().items()}}
    assert not torch.equal(src.nerf.sigma.weight, dst.nerf.sigma.weight)
    data.load_ckpt(dst, ckpt, "anim_nerf", prefixes_to_ignore=["body_model"])
    for (k, a), (_, b) in zip(src.nerf.state_dict().items(), dst.nerf.state_dict().items()):
        assert torch.equal(a, b), k
    rng = np.random.RandomState(2)
    with tempfile.TemporaryDirectory() as root:
        os.makedirs(os.path.join(root, "smpls"))
        os.makedirs(os.path.join(root, "cam000"))
        frame = {"betas": rng.randn(10), "global_orient": rng.randn(3), "body_pose": rng.randn(69), "transl": rng.randn(3),
                 "v_personal": np.zeros((4, 3)), "model_type": "smpl", "gender": "male"}
        data.write_pickle_file(os.path.join(root, "smpls", "000007.pkl"), frame)
        pts = rng.rand(500, 3)
        dist = rng.randn(500) * 0.1
        tmpl = {"betas": rng.randn(10), "body_pose": rng.randn(69), "global_orient": np.zeros(3), "transl": np.zeros(3),
                "points": pts, "distances": dist, "model_type": "smpl", "gender": "male"}
        data.write_pickle_file(os.path.join(root, "smpl_template.pkl"), tmpl)
        data.write_pickle_file(os.path.join(root, "cam000", "camera.pkl"), _cam(golden("formats")))
        p = data.load_body_model_params(root, 7)
        assert set(p) == {"betas", "global_orient", "body_pose", "transl"} and p["body_pose"].shape == (69,)
        assert p["betas"].dtype == torch.float32 and np.allclose(p["transl"].numpy(), frame["transl"].astype(np.float32))
        t, fg, bg = data.load_template(root)
        assert set(t) == {"betas_template", "global_orient_template", "body_pose_template", "transl_template"}
        assert fg.shape[0] == (dist < -0.02).sum() and bg.shape[0] == (dist > 0.10).sum()
        f, b = data.sample_prior_points(fg, bg, 128, generator=torch.Generator().manual_seed(0))
        assert f.shape == (128, 3) and b.shape == (128, 3)
        assert data.load_camera(root, 0)["width"] == 24
    assert data.frame_index([1, 5, 9]) == {1: 0, 5: 1, 9: 2}


def test_orbit_is_rodrigues_of_the_reference_axis_angles():
    from scipy.spatial.transform import Rotation
    from anim_nerf_amd import data
    P = data.orbit_transforms(n_views=8, angle=20.0).numpy()
    assert P.shape == (8, 4, 4)
    rx = Rotation.from_rotvec([-np.radians(20.0), 0, 0]).as_matrix()              # cv2.Rodrigues(axis-angle)
    for i in range(8):
        ry = Rotation.from_rotvec([0, 2 * np.pi * i / 8, 0]).as_matrix()
        assert np.allclose(P[i, :3, :3], ry @ rx, atol=1e-6) and np.allclose(P[i, 3], [0, 0, 0, 1])
        assert np.allclose(P[i, :3, 3], 0)


def test_mocap_sequence_matches_reference(tmp_path):
    """data.load_mixamo_smpl against the reference's (novel_pose.py:26-41, tests/golden/mocap.npz): frame selection, the pose
    split and the root translation (cam[1], cam[2], 0)."""
    import pickle
    from anim_nerf_amd import data
    g = golden("mocap")
    os.makedirs(tmp_path / "0007")
    with open(tmp_path / "0007" / "result.pkl", "wb") as f:
        pickle.dump({"anim_len": int(g["anim_len"]), "smpl_array": g["smpl_array"], "cam_array": g["cam_array"]}, f)
    mocap = data.load_mixamo_smpl(str(tmp_path), "0007", int(g["skip"]))
    assert len(mocap) == len(g["transl"])
    for i, m in enumerate(mocap):
        for k in ("global_orient", "body_pose", "transl", "cam"):
            assert np.array_equal(np.asarray(m[k], dtype=np.float64), np.asarray(g[k][i], dtype=np.float64)), (i, k)
