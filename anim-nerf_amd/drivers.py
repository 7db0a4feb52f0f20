"""Runnable drivers over the rendering path: the novel-view orbit (novel_view.py:144-210) and mesh extraction — sigma grid,
marching cubes, rescale, .obj (extract_mesh.py:142-173).  Everything heavy is `batched_inference(P=...)` / `sigma_grid`; this file
is the argument handling, the data-folder / checkpoint plumbing (`anim_nerf_amd.data`) and the image writer the
reference delegates to torchvision / imageio (absent here: PNGs are written with zlib).

    python -m anim_nerf_amd.drivers novel_view   --synthetic --n_views 8 --img_wh 256 256 --out out/nv
    python -m anim_nerf_amd.drivers novel_view   --root_dir DATA --ckpt_path CKPT --frame_id 1 --n_views 120
    python -m anim_nerf_amd.drivers extract_grid --synthetic --N_grid 256 --out out/mesh

`extract_grid` writes the thresholded sigma volume (`sigma.npy`), the posed SMPL mesh (`smpl.obj`) and the level-set mesh
(`mesh.obj`: `anim_nerf_amd.mesh`, marching cubes on the GPU — PyMCubes is absent from the image, so that step is held to
properties, not to PyMCubes' output).  The GIF writer is out of scope.
"""
from __future__ import annotations

import argparse
import os
import struct
import time
import zlib

import numpy as np
import torch

from . import data, synthetic
from .anim_nerf import AnimNeRF
from .render import batched_inference, sigma_grid
from .volume_rendering import VolumeRenderer


# ----------------------------------------------------------------------------------------------- image output
def write_png(path: str, img: np.ndarray):
    """img[H,W,C] uint8, C in {1,3,4} -> PNG (zlib only)."""
    h, w, c = img.shape
    colour = {1: 0, 3: 2, 4: 6}[c]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, colour, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def depth_image(depth: torch.Tensor) -> np.ndarray:
    """utils/__init__.py visualize_depth without the cv2 colour map: depth normalised to [0,1] over the image, grey."""
    x = torch.nan_to_num(depth.float())
    lo, hi = x.min(), x.max()
    x = (x - lo) / (hi - lo + 1e-8)
    return (x.clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy()[..., None]


# ----------------------------------------------------------------------------------------------- scene set-up
def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("the rendering path runs on a GPU only (libanimnerf_hip.so, no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _synthetic_scene(args, dev):
    """seeded stand-in for a People-Snapshot folder + checkpoint (no dataset / checkpoint ships with this repository)"""
    tbl = synthetic.make_smpl_table(0)
    torch.manual_seed(0)
    model = AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True, use_fine=True,
                     dis_threshold=args.dis_threshold, mlp_mode=args.mlp_mode).eval().to(dev)
    with torch.no_grad():                                   # random-init sigma is flat: spread it so there is a surface
        g = torch.Generator().manual_seed(5)
        probe = (torch.rand(1, 4096, 3, generator=g) * 1.6 - 0.8).to(dev)
        for net in (model.nerf, model.nerf_fine):
            mode, net.mlp_mode = net.mlp_mode, "f32"
            med = net(probe)[1].median().item()
            net.mlp_mode = mode
            net.sigma.weight.mul_(3000.0)
            net.sigma.bias.mul_(3000.0).add_(-3000.0 * med)
    W, H = args.img_wh
    c2w, focal, cen = synthetic.pinhole_camera(H, W)
    from .rays import gen_rays
    rays = gen_rays(torch.from_numpy(c2w).to(dev), H, W, focal.tolist(), 0.1, 10.0, cen.tolist())
    pose = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.animated_pose_params(seed=100 + args.frame_id).items()}
    templ = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.template_pose_params().items()}
    return model, VolumeRenderer(n_coarse=args.n_coarse, n_fine=args.n_fine), rays, pose, templ


def _disk_scene(args, dev):
    """novel_view.py:19-76,144-190: checkpoint -> modules, camera.pkl -> rays, smpls/*.pkl + template -> parameters."""
    hp = data.load_hparams(args.ckpt_path)
    model = AnimNeRF(model_path=getattr(hp, "model_path", "smplx/models"), model_type=hp.model_type, gender=hp.gender,
                     freqs_xyz=hp.freqs_xyz, freqs_dir=hp.freqs_dir, use_view=hp.use_view, use_unpose=hp.use_unpose,
                     k_neigh=hp.k_neigh, use_knn=hp.use_knn, use_fine=hp.n_importance > 0, share_fine=hp.share_fine,
                     dis_threshold=args.dis_threshold, query_inside=getattr(hp, "query_inside", False),
                     mlp_mode=args.mlp_mode).eval()
    data.load_ckpt(model, args.ckpt_path, model_name="anim_nerf")
    model = model.to(dev)
    vr = VolumeRenderer(n_coarse=hp.n_samples, n_fine=hp.n_importance, n_fine_depth=getattr(hp, "n_depth", 0),
                        share_fine=hp.share_fine, white_bkgd=getattr(hp, "white_bkgd", True))
    cam = data.load_camera(hp.root_dir if args.root_dir is None else args.root_dir, args.cam_id)
    img_wh = tuple(args.img_wh) if args.img_wh else tuple(hp.img_wh)
    cam = data.rescale_camera(cam, img_wh)
    args.img_wh = img_wh
    rays = data.camera_rays(cam, getattr(hp, "near", 0.1), getattr(hp, "far", 10.0), device=dev)
    root = hp.root_dir if args.root_dir is None else args.root_dir
    pose = {k: v[None].to(dev) for k, v in data.load_body_model_params(root, args.frame_id, hp.model_type).items()}
    templ_params, _, _ = data.load_template(root, hp.model_type)
    templ = {k.replace("_template", ""): v[None].to(dev) for k, v in templ_params.items()}
    if args.template:
        pose["body_pose"] = templ["body_pose"]
    return model, vr, rays, pose, templ


def _scene(args):
    dev = _device()
    if args.synthetic:
        return _synthetic_scene(args, dev)
    if not args.ckpt_path:
        raise SystemExit("give --ckpt_path (and --root_dir), or --synthetic")
    return _disk_scene(args, dev)


# ----------------------------------------------------------------------------------------------- drivers
def novel_view(args):
    """novel_view.py:144-210: n_views renders of one frame on an orbit about the body's vertical axis."""
    model, vr, rays, pose, templ = _scene(args)
    W, H = args.img_wh
    pose = dict(pose)
    pose["betas"] = pose["betas"].clone()
    pose["betas"][:, 1] += args.betas_2th
    os.makedirs(os.path.join(args.out, "images"), exist_ok=True)
    os.makedirs(os.path.join(args.out, "depths"), exist_ok=True)
    flat = rays.view(1, H * W, -1)
    P_all = data.orbit_transforms(args.n_views, args.angle).to(flat.device)
    t0 = time.perf_counter()
    for i in range(args.n_views):
        out = batched_inference(vr, model, flat, pose, templ, P=P_all[i][None, None], chunk=args.chunk)
        tag = "_fine" if "rgbs_fine" in out else ""
        img = (out["rgbs" + tag].view(H, W, 3).clamp(0, 1) * 255).round().to(torch.uint8)
        mask = (out["alphas" + tag].view(H, W, 1).clamp(0, 1) * 255).round().to(torch.uint8)
        write_png(os.path.join(args.out, "images", f"{i:06d}.png"), torch.cat([img, mask], -1).cpu().numpy())
        write_png(os.path.join(args.out, "depths", f"{i:06d}.png"), depth_image(out["depths" + tag].view(H, W)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.n_views} views of {W}x{H} in {dt:.2f} s ({args.n_views * W * H / dt / 1e6:.2f} M rays/s incl. PNG output) -> {args.out}")
    return args.out


def novel_pose(args):
    """novel_pose.py:118-176: one camera, the subject driven through a motion-capture sequence — per frame (global_orient,
    body_pose) from the sequence, the subject's own betas, transl = the subject's transl + the sequence's root translation —
    rendered with `batched_inference`; images (RGBA), masks and depth maps as PNG.  The sequence: `--actions_dir/--action_type`
    (load_mixamo_smpl), or with --synthetic a seeded walk-like swing of the limbs (no mocap file ships with this repository).
    Not reproduced: the SMPL overlay renders (pyrender) and the GIF (imageio) — neither library is in the image."""
    model, vr, rays, pose, templ = _scene(args)
    W, H = args.img_wh
    dev = rays.device
    if args.synthetic:
        n = args.n_frames
        base = synthetic.animated_pose_params(seed=300)
        t = np.linspace(0, 2 * np.pi, n, endpoint=False, dtype=np.float32)
        swing = np.zeros((n, 69), dtype=np.float32)
        swing[:, 0], swing[:, 3] = 0.5 * np.sin(t), -0.5 * np.sin(t)                # hips, x axis
        swing[:, 47], swing[:, 50] = 0.4 * np.sin(t), -0.4 * np.sin(t)              # shoulders
        mocap = data.mocap_frames({"anim_len": n, "smpl_array": np.concatenate([np.tile(base["global_orient"], (n, 1)), base["body_pose"] * 0.3 + swing], 1),
                                   "cam_array": np.stack([np.ones(n), 0.02 * np.sin(t), 0.01 * np.cos(2 * t)], 1)}, args.frame_skip)
    else:
        mocap = data.load_mixamo_smpl(args.actions_dir, args.action_type, args.frame_skip)
    for sub in ("images", "masks", "depths"):
        os.makedirs(os.path.join(args.out, sub), exist_ok=True)
    flat = rays.view(1, H * W, -1)
    t0 = time.perf_counter()
    for i, fr in enumerate(mocap):
        params = {"betas": pose["betas"], "global_orient": torch.from_numpy(fr["global_orient"]).float()[None].to(dev),
                  "body_pose": torch.from_numpy(fr["body_pose"][:69]).float()[None].to(dev),
                  "transl": pose["transl"] + torch.from_numpy(fr["transl"]).float()[None].to(dev)}
        out = batched_inference(vr, model, flat, params, templ, chunk=args.chunk)
        tag = "_fine" if "rgbs_fine" in out else ""
        img = (out["rgbs" + tag].view(H, W, 3).clamp(0, 1) * 255).round().to(torch.uint8)
        mask = (out["alphas" + tag].view(H, W, 1).clamp(0, 1) * 255).round().to(torch.uint8)
        write_png(os.path.join(args.out, "images", f"{i:06d}.png"), torch.cat([img, mask], -1).cpu().numpy())
        write_png(os.path.join(args.out, "masks", f"{i:06d}.png"), mask.expand(-1, -1, 3).contiguous().cpu().numpy())
        write_png(os.path.join(args.out, "depths", f"{i:06d}.png"), depth_image(out["depths" + tag].view(H, W)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{len(mocap)} poses of {W}x{H} in {dt:.2f} s ({len(mocap) * W * H / dt / 1e6:.2f} M rays/s incl. PNG output) -> {args.out}")
    return args.out


def extract_grid(args):
    """extract_mesh.py:142-173: posed SMPL mesh (smpl.obj), the thresholded sigma volume, and the level-set mesh (mesh.obj) —
    sigma grid, marching cubes (anim_nerf_amd.mesh: PyMCubes is not in the image, parity-unpinned), rescale, export."""
    model, _, rays, pose, templ = _scene(args)
    with torch.no_grad():
        model.set_body_model(pose, templ)
        model.convert_to_body_model_space(rays.view(1, -1, rays.shape[-1])[:, :1])
        model.clac_ober2cano_transform()
        t0 = time.perf_counter()
        sig, _ = sigma_grid(model, args.N_grid, args.x_range, args.y_range, args.z_range, chunk=args.chunk_points)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    os.makedirs(args.out, exist_ok=True)
    N = args.N_grid
    vol = sig.view(N, N, N).cpu().numpy() - args.sigma_threshold          # np.maximum(sigmas, 0) - threshold (:159-160)
    np.save(os.path.join(args.out, "sigma.npy"), vol.astype(np.float32))
    center = ((model.verts.max(dim=1)[0] + model.verts.min(dim=1)[0]) / 2.)[0].cpu().numpy()
    np.save(os.path.join(args.out, "center.npy"), center)
    verts = model.verts[0].cpu().numpy()
    faces = np.asarray(model.body_model.faces).reshape(-1, 3)
    with open(os.path.join(args.out, "smpl.obj"), "w") as f:               # mcubes.export_obj layout (:147-149)
        for v in verts:
            f.write(f"v {v[0]} {v[1]} {v[2]}\n")
        for t in faces:
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
    occupied = int((vol > 0).sum())
    print(f"{N}^3 sigma grid in {dt * 1e3:.1f} ms ({N ** 3 / dt / 1e9:.2f} G points/s), {occupied} voxels above the threshold -> {args.out}")
    # extract_mesh.py:162-173: (smooth,) marching cubes on -sigmas at level 0, index coordinates -> world, + centre, mesh.obj
    from . import mesh
    field = -(sig.view(N, N, N) - args.sigma_threshold)
    if getattr(args, "smooth", False):
        field = mesh.gaussian_smooth(field)
    t0 = time.perf_counter()
    v_idx, faces_m = mesh.marching_cubes(field.contiguous(), 0.0)
    torch.cuda.synchronize()
    dt_m = time.perf_counter() - t0
    verts_m = mesh.mcubes_to_world(v_idx.cpu().numpy(), N, args.x_range, args.y_range, args.z_range) + center
    mesh.export_obj(verts_m, faces_m.cpu().numpy(), os.path.join(args.out, "mesh.obj"))
    print(f"marching cubes: {verts_m.shape[0]} vertices, {faces_m.shape[0]} triangles in {dt_m * 1e3:.1f} ms -> {os.path.join(args.out, 'mesh.obj')}")
    return args.out


def parser():
    ap = argparse.ArgumentParser(prog="python -m anim_nerf_amd.drivers")
    sub = ap.add_subparsers(dest="cmd", required=True)

    def common(p):
        p.add_argument("--ckpt_path", type=str, default=None, help="Lightning checkpoint of the reference's AnimNeRFSystem")
        p.add_argument("--root_dir", type=str, default=None, help="data folder (default: the checkpoint's hparams.root_dir)")
        p.add_argument("--synthetic", action="store_true", help="seeded synthetic body / camera / weights instead of files")
        p.add_argument("--frame_id", type=int, default=1)
        p.add_argument("--cam_id", type=int, default=0)
        p.add_argument("--template", action="store_true", help="render the template pose")
        p.add_argument("--dis_threshold", type=float, default=0.2)
        p.add_argument("--img_wh", type=int, nargs=2, default=None)
        p.add_argument("--n_coarse", type=int, default=64)
        p.add_argument("--n_fine", type=int, default=64)
        p.add_argument("--mlp_mode", default="bf16", choices=["bf16", "f32"])
        p.add_argument("--out", type=str, required=True)
    nv = sub.add_parser("novel_view")
    common(nv)
    nv.add_argument("--chunk", type=int, default=1 << 20)
    nv.add_argument("--betas_2th", type=float, default=0.0)
    nv.add_argument("--n_views", type=int, default=120)
    nv.add_argument("--angle", type=int, default=0)
    npz = sub.add_parser("novel_pose", help="novel_pose.py: the subject driven through a motion-capture sequence")
    common(npz)
    npz.add_argument("--chunk", type=int, default=1 << 20)
    npz.add_argument("--actions_dir", type=str, default="mocap/mixamo/")
    npz.add_argument("--action_type", type=str, default="0007")
    npz.add_argument("--frame_skip", type=int, default=2)
    npz.add_argument("--n_frames", type=int, default=16, help="--synthetic: frames of the seeded sequence (before frame_skip)")

    eg = sub.add_parser("extract_grid")
    common(eg)
    eg.add_argument("--N_grid", type=int, default=256)
    eg.add_argument("--x_range", type=float, nargs=2, default=[-1.2, 1.2])
    eg.add_argument("--y_range", type=float, nargs=2, default=[-1.2, 1.2])
    eg.add_argument("--z_range", type=float, nargs=2, default=[-1.2, 1.2])
    eg.add_argument("--sigma_threshold", type=float, default=20.0)
    eg.add_argument("--chunk_points", type=int, default=1 << 24)
    eg.add_argument("--smooth", action="store_true", help="Gaussian-filter the volume first (the reference: mcubes.smooth)")
    return ap


def main(argv=None):
    args = parser().parse_args(argv)
    if args.synthetic and args.img_wh is None:
        args.img_wh = [256, 256]
    return {"novel_view": novel_view, "novel_pose": novel_pose, "extract_grid": extract_grid}[args.cmd](args)


if __name__ == "__main__":
    main()
