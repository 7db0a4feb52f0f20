#!/usr/bin/env python3
"""rays/s of the Anim-NeRF rendering hot path on MI355X.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One step = one full 1024 x 1024 frame per GPU through the whole path (per-frame SMPL state, rays to
the body frame, 64 coarse samples -> coarse net -> composite -> 64 importance samples -> fine net on
all 128 sorted samples -> composite).  Workload = BASELINE.json configs[1] (no LBS warp) by default,
`--workload cfg3` adds the inverse-LBS / 4-NN warp.  Rays are resident in HBM before the timed region.
Ranks render independent frames (no data-path collective): weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import math
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MLP_FLOP_PER_POINT = 1_179_904          # 589,952 MACs, full rgb + sigma evaluation
PEAK_BF16_TFLOPS = 2500.0               # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3                 # f32-input MFMA = fp32 vector peak


def _psnr(a, b):
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="cfg2/cfg3: BASELINE configs[1]/[2] (inference); cfg4: configs[3] training step (64+32, 32x32 rays per frame); "
                         "cfg5: configs[4] 512^3 sigma grid, voxel-sharded")
    ap.add_argument("--frames-per-gpu", type=int, default=16, help="cfg4: frames (of 1024 rays) per step per GPU")
    ap.add_argument("--mode", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--hw", type=int, default=1024)
    ap.add_argument("--n-coarse", type=int, default=64)
    ap.add_argument("--n-fine", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=1 << 20, help="rays per renderer call")
    ap.add_argument("--cpu-rays", type=int, default=16384, help="upper bound on the rays timed on the host oracle (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target duration of the host-oracle sample")
    ap.add_argument("--sigma-gain", type=float, default=3000.0,
                    help="scale the sigma heads about their median (0 = literal random init, which renders a blank image)")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--dense", action="store_true",
                    help="cfg3/cfg4: run the MLP on every sample as the reference does, also on those outside dis_threshold "
                         "(sigma = -1e5, composite weight 0); default: only on the valid ones — same image, bit for bit")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    import anim_nerf_amd as ana
    from anim_nerf_amd import ops, synthetic as syn

    ana._lib.load()                                   # fails loudly if the HIP library is missing
    if args.workload == "cfg4":
        return train_bench(args, rank, local_rank, world, dev)
    if args.workload == "cfg5":
        return grid_bench(args, rank, local_rank, world, dev)
    use_warp = args.workload == "cfg3"
    tbl = syn.make_smpl_table(0)
    torch.manual_seed(0)
    model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=use_warp, use_knn=True,
                         use_fine=True, mlp_mode=args.mode).eval().to(dev)
    if args.sigma_gain > 0:
        # random-init sigma is ~0.017 +- 0.003 (one sign everywhere): every ray renders white or saturates on the far
        # sample.  Spread it about its median so that compositing, importance sampling and the PSNR mean something.
        # Same FLOPs, same kernels; only the numbers in the sigma row change.
        g = torch.Generator().manual_seed(5)
        probe = (torch.rand(1, 4096, 3, generator=g) * 1.6 - 0.8).to(dev)
        with torch.no_grad():
            for net in (model.nerf, model.nerf_fine):
                net.mlp_mode = "f32"
                med = net(probe)[1].median().item()
                net.mlp_mode = args.mode
                net.sigma.weight.mul_(args.sigma_gain)
                net.sigma.bias.mul_(args.sigma_gain).add_(-args.sigma_gain * med)
    model.skip_invalid_samples = not args.dense
    vr = ana.VolumeRenderer(n_coarse=args.n_coarse, n_fine=args.n_fine, white_bkgd=True)
    H = W = args.hw
    c2w, focal, cen = syn.pinhole_camera(H, W)
    # each rank renders its own frame: same camera, rank-seeded pose for the warp workload
    pose_np = syn.animated_pose_params(seed=100 + rank) if use_warp else syn.static_pose_params()
    pose = {k: torch.from_numpy(v).to(dev) for k, v in pose_np.items()}
    templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), H, W, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
    n_rays = rays.shape[1]

    def step():
        return ana.batched_inference(vr, model, rays, pose, templ, chunk=args.chunk)

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    barrier()
    ops.KERNEL_TIMING = []                            # (name, start_event, end_event, units) per launch
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    timing, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None
    elapsed = ana.max_over_ranks(elapsed, dev)

    # ---- dominant kernel: fused MLP; HIP events were recorded on the launch stream inside the timed region
    per_kernel = {}
    for name, e0, e1, units in timing:
        d = per_kernel.setdefault(name, [0.0, 0, 0])
        d[0] += e0.elapsed_time(e1) * 1e-3
        d[1] += int(units)                            # (a device counter where the MLP ran on a compacted list)
        d[2] += 1
    mlp_s, mlp_pts, mlp_launches = per_kernel.get("mlp_forward", [0.0, 0, 0])
    peak = PEAK_BF16_TFLOPS if args.mode == "bf16" else PEAK_F32_TFLOPS
    achieved = (mlp_pts * MLP_FLOP_PER_POINT / mlp_s / 1e12) if mlp_s > 0 else 0.0
    kernel_share = {k: round(v[0] / elapsed * 1.0, 4) for k, v in per_kernel.items()}

    # HBM bytes per launch of the MLP kernel: bytes per point measured with rocprofv3 PMC passes (FETCH_SIZE and
    # WRITE_SIZE in separate runs, gfx950 correction applied; profiles/r01/mlp_hbm_traffic.json) x points per launch
    traffic = None
    tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "mlp_hbm_traffic.json")
    if args.mode == "bf16" and mlp_launches and os.path.exists(tf):
        with open(tf) as fh:
            traffic = json.load(fh)["bytes_per_point_indexed" if model.evaluate_valid_only else
                                    "bytes_per_point_explicit" if use_warp else "bytes_per_point"] * mlp_pts / mlp_launches

    result = {
        "metric": "rays/sec (64+64 samples, 256-wide MLP)",
        "value": n_rays * args.steps * world / elapsed,
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.mode,
        "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[1]: 1024x1024 render, 64 coarse + 64 fine, random-init 8x256 MLP x2"
                         + (" (sigma rows rescaled about their median)" if args.sigma_gain > 0 else "")
                         + ", fixed SMPL pose, no LBS warp" if not use_warp else
                         "BASELINE configs[2]: configs[1] + inverse-LBS / exact 4-NN canonical warp (V=6890, animated pose)"),
            "rays_per_step_per_gpu": n_rays, "n_coarse": args.n_coarse, "n_fine": args.n_fine,
            "mlp_evals_per_ray": args.n_coarse + (args.n_coarse + args.n_fine if args.n_fine else 0),
            "chunk_rays": args.chunk, "sharding": f"{world} independent frames (ray-parallel, no collective)",
            # warp on: samples farther than dis_threshold from the body are sigma = -1e5 / weight 0 whatever the MLP
            # says; the MLP runs on the others only (identical image; --dense evaluates all of them like the reference)
            "mlp_on_valid_samples_only": bool(model.evaluate_valid_only),
            "mlp_points_per_step": mlp_pts // max(args.steps, 1),
        },
        "roofline": {
            "kernel": f"mlp_kernel<{args.mode}> (fused Fourier encoding + 11 GEMMs)",
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak, "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC, profiles/r01/mlp_hbm_traffic.json: 20.6 B per point measured on the no-warp path / 20 algorithmic; 40.1 / 36 per evaluated point with the warp on)",
            "launches": mlp_launches, "avg_launch_ms": (mlp_s / mlp_launches * 1e3) if mlp_launches else None,
            "flop_per_point": MLP_FLOP_PER_POINT,
        },
        "kernel_time_share": kernel_share,
    }
    # everything around the MLP (sampling, point generation / warp, compaction, compositing,
    # importance sampling) against the HBM roofline on its COMPULSORY bytes: 52 B per ray + 36 B per sample
    # (z 4 B, canonical point 16 B, rgb-sigma 16 B).  With the warp on the binding resource is VALU issue (exact KNN),
    # so the HBM fraction there is a lower bound on how far those kernels are from their own limit.
    other_s = sum(v[0] for k, v in per_kernel.items() if k != "mlp_forward")
    evals = args.n_coarse + (args.n_coarse + args.n_fine if args.n_fine else 0)
    comp_bytes = n_rays * args.steps * (52 + 36 * evals)
    result["roofline_hbm_kernels"] = {
        "kernels": sorted(k for k in per_kernel if k != "mlp_forward"), "bound": "hbm" if not use_warp else "valu (exact 4-NN)",
        "achieved": comp_bytes / other_s / 1e9 if other_s > 0 else None, "peak": 8000.0, "unit": "GB/s",
        "frac": comp_bytes / other_s / 8e12 if other_s > 0 else None, "compulsory_bytes_per_ray": 52 + 36 * evals,
        "ms_per_step": other_s / args.steps * 1e3,
    }

    if rank == 0 and world == 1:
        if not args.no_psnr:
            # PSNR of this mode's image vs the fp32 parity path (pinned to the reference) on a centre crop
            c = min(256, H)
            lo = (H - c) // 2
            idx = (torch.arange(lo, lo + c)[:, None] * W + torch.arange(lo, lo + c)[None]).reshape(-1).to(dev)
            for net in (model.nerf, model.nerf_fine):
                net.mlp_mode = "f32"
            ref = ana.batched_inference(vr, model, rays[:, idx].contiguous(), pose, templ, chunk=1 << 16)
            for net in (model.nerf, model.nerf_fine):
                net.mlp_mode = args.mode
            result["psnr_vs_fp32_path_db"] = _psnr(out["rgbs_fine"][:, idx].cpu(), ref["rgbs_fine"].cpu())
        if args.cpu_rays > 0:
            result["cpu_baseline"], pick, ref = cpu_baseline(args, tbl, model, rays, pose_np, use_warp)
            # the oracle's output as the checker: the same rays through the HIP path (fp32 parity mode and the benchmarked mode)
            sub = rays[:, pick.to(dev)].contiguous()
            key = "rgbs_fine" if args.n_fine else "rgbs"
            check = {"rays": int(pick.numel())}
            for mode in dict.fromkeys(("f32", args.mode)):
                for net in (model.nerf, model.nerf_fine):
                    net.mlp_mode = mode
                got = ana.batched_inference(vr, model, sub, pose, templ, chunk=1 << 16)[key].cpu()
                err = (got - ref[key]).abs().max(-1).values / ref[key].abs().max(-1).values.clamp_min(1e-3)
                check[mode] = {"max_rel_err_rgb": err.max().item(), "rays_within_1e-4": (err <= 1e-4).float().mean().item(),
                               "psnr_db": _psnr(got, ref[key])}
            for net in (model.nerf, model.nerf_fine):
                net.mlp_mode = args.mode
            result["oracle_check"] = check
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def grid_bench(args, rank, local_rank, world, dev):
    """BASELINE configs[4]: relu(sigma) of the fine field on a 512^3 grid around the posed body (extract_mesh.py:152-158),
    voxel-sharded: rank r evaluates slab r of the SAME grid (strong scaling), no collective on the data path."""
    import torch.distributed as dist
    import anim_nerf_amd as ana
    from anim_nerf_amd import ops, synthetic as syn
    N = 512
    tbl = syn.make_smpl_table(0)
    torch.manual_seed(0)
    model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True,
                         use_fine=True, mlp_mode=args.mode).eval().to(dev)
    g = torch.Generator().manual_seed(5)
    probe = (torch.rand(1, 4096, 3, generator=g) * 1.6 - 0.8).to(dev)
    with torch.no_grad():
        for net in (model.nerf, model.nerf_fine):
            net.mlp_mode = "f32"
            med = net(probe)[1].median().item()
            net.mlp_mode = args.mode
            net.sigma.weight.mul_(args.sigma_gain or 1.0)
            net.sigma.bias.mul_(args.sigma_gain or 1.0).add_(-(args.sigma_gain or 1.0) * med)
        model.skip_invalid_samples = not args.dense
        pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=100).items()}
        templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
        rays = torch.zeros(1, 1, 8, device=dev)
        rays[..., 5], rays[..., 7] = -1, 10
        model.set_body_model(pose, templ)
        model.convert_to_body_model_space(rays)
        model.clac_ober2cano_transform()

        def step():
            return ana.sigma_grid(model, N, chunk=1 << 25, rank=rank, world=world)

        def barrier():
            if world > 1:
                dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize(dev)
        for _ in range(args.warmup):
            step()
        barrier()
        ops.KERNEL_TIMING = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sig, _ = step()
        barrier()
        elapsed = ana.max_over_ranks(time.perf_counter() - t0, dev)
        timing, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None
    mlp_s = sum(e0.elapsed_time(e1) for n, e0, e1, u in timing if n == "mlp_forward") * 1e-3
    mlp_pts = sum(int(u) for n, e0, e1, u in timing if n == "mlp_forward")
    peak = PEAK_BF16_TFLOPS if args.mode == "bf16" else PEAK_F32_TFLOPS
    flop = 982_528                                            # sigma-only: trunk + sigma row
    achieved = mlp_pts * flop / mlp_s / 1e12 if mlp_s else 0.0
    result = {
        "metric": "grid points/sec, 512^3 sigma query (mesh extraction input)", "value": N ** 3 * args.steps / elapsed,
        "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.mode, "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: 512^3 sigma grid around the posed body, fine network, voxel-sharded "
                               "(contiguous slabs, no collective)", "grid": N, "mlp_on_valid_voxels_only": bool(model.evaluate_valid_only),
                   "mlp_points_per_step_this_rank": mlp_pts // max(args.steps, 1), "occupied_voxels_this_rank": int((sig > 0).sum())},
        "roofline": {"kernel": f"mlp_kernel<{args.mode}, sigma only>", "bound": "mfma", "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None, "flop_per_point": flop},
    }
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def train_bench(args, rank, local_rank, world, dev):
    """BASELINE configs[3] shape: one optimisation step = `frames_per_gpu` frames x 32x32 rays, 64 coarse + 32 fine,
    perturb = 1, rgb/alpha/foreground/background losses, backward, ONE flat gradient all-reduce (RCCL), Adam."""
    import torch.distributed as dist
    import anim_nerf_amd as ana
    from anim_nerf_amd import ops, synthetic as syn
    tbl = syn.make_smpl_table(0)
    torch.manual_seed(0)
    model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True,
                         use_fine=True, mlp_mode=args.mode).to(dev)
    model.skip_invalid_samples = not args.dense
    hp = ana.TrainHParams(n_samples=64, n_importance=32, chunk=2048)
    F = args.frames_per_gpu
    table = ana.BodyModelParams(114).to(dev)                  # 114 training frames (configs/people_snapshot/male-3-casual.yaml)
    seeded = syn.animated_pose_params(seed=200 + rank, bs=114)
    for name in table.param_names:                            # optim_body_params: True
        table.init_parameters(name, torch.from_numpy(seeded[name]).to(dev), requires_grad=True)
    trainer = ana.Trainer(model, ana.VolumeRenderer(n_coarse=64, n_fine=32), hp, body_model_params=table)
    frame_idx = torch.arange(F, device=dev) * (114 // F)
    c2w, focal, cen = syn.pinhole_camera(32, 32)
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 32, 32, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(F, 1, 1, 1)
    pose = None                                               # looked up from the table each step
    templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
    g = torch.Generator().manual_seed(rank)
    rgbs = torch.rand(F, 32, 32, 3, generator=g).to(dev)
    alphas = (torch.rand(F, 32, 32, 1, generator=g) > 0.5).float().to(dev)
    fg = (torch.rand(F, 128, 3, generator=g) * 0.4 - 0.2).to(dev)
    bg = (torch.rand(F, 128, 3, generator=g) * 2 - 1).to(dev)

    def step():
        return trainer.step(rays, rgbs, alphas, pose, templ, fg, bg, perturb=1.0, frame_idx=frame_idx)

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    ops.KERNEL_TIMING = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    barrier()
    elapsed = ana.max_over_ranks(time.perf_counter() - t0, dev)
    timing, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None
    mlp_s = sum(e0.elapsed_time(e1) for n, e0, e1, u in timing if n == "mlp_forward_save") * 1e-3
    mlp_pts = sum(int(u) for n, e0, e1, u in timing if n == "mlp_forward_save")
    peak = PEAK_BF16_TFLOPS if args.mode == "bf16" else PEAK_F32_TFLOPS
    achieved = mlp_pts * MLP_FLOP_PER_POINT / mlp_s / 1e12 if mlp_s else 0.0
    n_rays = F * 1024
    result = {
        "metric": "rays/sec, training step (64+32 samples, 256-wide MLP x2, fwd+bwd+Adam)",
        "value": n_rays * args.steps * world / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.mode, "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] shape: train step, %d frames x 32x32 rays per GPU, 64 coarse + 32 fine, "
                               "perturb=1, rgb + alpha + fg/bg + normals losses (reference defaults), pose refinement on (optim_body_params), "
                               "flat-gradient all-reduce (4.7 MB) + Adam" % F,
                   "mlp_on_valid_samples_only": bool(model.evaluate_valid_only),
                   "mlp_rows_per_step": mlp_pts // max(args.steps, 1),
                   "rays_per_step_per_gpu": n_rays, "grad_floats": sum(p.numel() for p in trainer.params)},
        "roofline": {"kernel": f"mlp_kernel<{args.mode}, save> (training forward)", "bound": "mfma", "achieved": achieved,
                     "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None},
        "final_loss": float(loss),
    }
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, tbl, model, rays, pose_np, use_warp):
    """The oracle (CPU restatement of the reference, torch CPU ops, all host cores) on a bounded
    sample of the same workload: `cpu_rays` rays of the same frame, same sample counts."""
    from anim_nerf_amd import synthetic as syn
    from oracle import animnerf_oracle as orc
    import anim_nerf_amd as ana
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(cores)
    bm = ana.SMPL(data_struct=tbl)
    otbl = dict(v_template=bm.v_template, shapedirs=bm.shapedirs, posedirs=bm.posedirs, J_regressor=bm.J_regressor,
                parents=bm.parents, lbs_weights=bm.lbs_weights, extra_joints_idxs=bm.vertex_joint_selector.extra_joints_idxs)
    Pc = {k: v.detach().cpu() for k, v in model.nerf.named_parameters()}
    Pf = {k: v.detach().cpu() for k, v in model.nerf_fine.named_parameters()}
    pose = {k: torch.from_numpy(v) for k, v in pose_np.items()}
    templ = {k: torch.from_numpy(v) for k, v in syn.template_pose_params().items()}
    kw = dict(n_coarse=args.n_coarse, n_fine=args.n_fine, use_unpose=use_warp, chunk=512 if not use_warp else 128,
              knn_chunk=2048)
    # warm-up doubles as calibration: the thread count that serves this op mix best (hundreds of threads on ops this small
    # lose to a few dozen), then the sample size for about --cpu-seconds of work
    centre = rays.shape[1] // 2
    probe = rays[:, centre:centre + 64].cpu().contiguous()
    orc.render_frame(otbl, Pc, Pf, probe[:, :8], pose, templ, **kw)
    rate, best = 0.0, cores
    for t in sorted({min(cores, c) for c in (8, 16, 32, 64, cores)}):
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        orc.render_frame(otbl, Pc, Pf, probe, pose, templ, **kw)
        r = 64 / (time.perf_counter() - t0)
        if r > rate:
            rate, best = r, t
    torch.set_num_threads(best)
    n = int(min(args.cpu_rays, max(64, rate * args.cpu_seconds)))
    # a regular grid over the central half of the image (where the body is: the oracle's cost does not depend on the
    # content, the check below does)
    H = W = int(round(rays.shape[1] ** 0.5))
    side = max(2, int(n ** 0.5))
    ys = torch.linspace(H // 4, 3 * H // 4 - 1, side).long()
    xs = torch.linspace(W // 4, 3 * W // 4 - 1, side).long()
    pick = (ys[:, None] * W + xs[None, :]).reshape(-1)
    stride = f"{side}x{side} grid over the central half"
    sample = rays[:, pick.to(rays.device)].cpu().contiguous()
    t0 = time.perf_counter()
    ref = orc.render_frame(otbl, Pc, Pf, sample, pose, templ, **kw)
    dt = time.perf_counter() - t0
    base = {"value": sample.shape[1] / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "host_cores": cores, "kind": "port",
            "sample": f"{sample.shape[1]} rays of the same frame ({stride}), {args.n_coarse}+{args.n_fine} samples, "
                      f"oracle/animnerf_oracle.py on torch CPU fp32, {dt:.1f} s"}
    return base, pick, ref


if __name__ == "__main__":
    main()
