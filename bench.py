#!/usr/bin/env python3
"""rays/s of the Anim-NeRF rendering hot path on MI355X.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One step = one full 1024 x 1024 frame per GPU through the whole path (per-frame SMPL state, rays to
the body frame, 64 coarse samples -> coarse net -> composite -> 64 importance samples -> fine net on
all 128 sorted samples -> composite).  The headline line is BASELINE.json configs[1] (no LBS warp) in bf16;
rays are resident in HBM before the timed region.  Ranks render independent frames (no data-path collective):
weak scaling; `--scaling strong` slices ONE frame's rays over the ranks instead (per-frame setup replicated).

The same JSON line also carries, each with its own warm-up and timed region (`--no-extras` skips them):
  "modes"     : {"f32": ...}                   the parity-grade arithmetic on the same workload
  "workloads" : {"cfg3", "cfg3_dense", "cfg4", "cfg5", "cfg2_strong"}   BASELINE configs[2], [3], [4]; strong scaling of [1]
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MLP_FLOP_PER_POINT = 1_179_904          # 589,952 MACs, full rgb + sigma evaluation
MLP_FLOP_SIGMA_ONLY = 982_528           # trunk + sigma row
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}    # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


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
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="cfg2/cfg3 at N > 1: weak = one frame per GPU; strong = one frame's rays sliced over the GPUs; "
                         "cfg4: weak = frames-per-gpu frames on every rank; strong = ONE batch of 16 frames split over the ranks "
                         "(the reference's DataParallel partitioning)")
    ap.add_argument("--frames-per-gpu", type=int, default=16, help="cfg4: frames (of 1024 rays) per step per GPU")
    ap.add_argument("--refine", action="store_true",
                    help="cfg4: the `_refine` stage of the shipped configs — networks frozen, only the poses train (train.py:433-437)")
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
    ap.add_argument("--no-extras", action="store_true", help="only the headline workload (no modes / workloads objects)")
    ap.add_argument("--one-pass", action="store_true",
                    help="cfg2: render through the one-pass ray-march kernel (anr_ray_march) instead of the staged launches")
    ap.add_argument("--extras-only", action="store_true",
                    help="(internal) the child job of an N > 1 run: the modes / workloads objects without the headline")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="N > 1 plumbing check without a GPU (tests/test_distributed_cpu.py): launch, rendezvous, verified "
                         "all-reduce, barrier-bracketed timing of a sleep, one JSON line")
    ap.add_argument("--dense", action="store_true",
                    help="cfg3/cfg4: run the MLP on every sample as the reference does, also on those outside dis_threshold "
                         "(sigma = -1e5, composite weight 0); default: only on the valid ones — same image, bit for bit")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks as a
    CHILD `torch.distributed.run` (the reference gets the same from Lightning's `gpus=-1`, train.py:451-458), relay its
    output and leave with its exit code.  Nothing in this process has touched the GPU when the child starts, and the child
    is a child, not an exec."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)     # stderr passes through
    line = None
    for text in proc.stdout:
        t = text.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t                                   # rank 0's one JSON line
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    raise SystemExit(rc if rc != 0 or line is not None else 1)


class Ctx:
    """process-group plumbing shared by every timed region"""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        # (test hooks: ANR_BENCH_BACKEND=gloo + ANR_BENCH_ONE_DEVICE=1 run the N > 1 code path on a single-GPU box)
        self.backend = os.environ.get("ANR_BENCH_BACKEND", "nccl")
        self.collective_ranks = 1
        self.rank_ms = None
        if args.plumbing_only and not (self.backend == "nccl" and torch.cuda.device_count() > self.local_rank):
            self.dev = torch.device("cpu")
        else:
            dev_index = 0 if os.environ.get("ANR_BENCH_ONE_DEVICE") else self.local_rank
            torch.cuda.set_device(dev_index)
            self.dev = torch.device("cuda", dev_index)
        if self.world > 1:
            import datetime
            import torch.distributed as dist
            # a rank that dies (OOM, HIP error on one GPU) must not leave the others in a collective for ever
            limit = datetime.timedelta(seconds=float(os.environ.get("ANR_BENCH_COLLECTIVE_TIMEOUT", "600")))
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev, timeout=limit)    # "nccl" = RCCL over xGMI on ROCm
            else:
                dist.init_process_group(self.backend, timeout=limit)
            # the number the line reports as `collective_ranks` comes out of a real all-reduce on the data backend
            one = torch.ones(1, dtype=torch.float32, device=self.dev)
            dist.all_reduce(one)
            self.collective_ranks = int(round(one.item()))
            if self.collective_ranks != self.world or dist.get_world_size() != self.world:
                raise SystemExit(f"all-reduce over {self.backend} summed {self.collective_ranks} ranks, expected {self.world}")

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            if self.backend == "nccl":
                dist.barrier(device_ids=[self.dev.index])
            else:
                dist.barrier()
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)

    def all_ok(self, ok):
        """True iff `ok` on every rank (one MIN all-reduce): a rank-local failure becomes everybody's failure instead of a
        hang in the next barrier."""
        if self.world == 1:
            return ok
        import torch.distributed as dist
        flag = torch.tensor([1.0 if ok else 0.0], device=self.dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return flag.item() > 0.5

    def spread(self, elapsed, steps):
        """per-rank ms per step: (fastest rank, slowest rank)"""
        if self.world == 1:
            return [elapsed / steps * 1e3] * 2
        import torch.distributed as dist
        t = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [-t[1].item() / steps * 1e3, t[0].item() / steps * 1e3]

    def timed(self, step, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks.
        -> (seconds, per-kernel HIP-event records, last output)"""
        import anim_nerf_amd as ana
        from anim_nerf_amd import ops
        out = None
        for _ in range(warmup):
            out = step()
        self.barrier()
        ops.KERNEL_TIMING = []                            # (name, start_event, end_event, units, bytes) per launch
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        # this rank's own clock stops when ITS queue is empty; the job's clock (below) after the barrier
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        own = time.perf_counter() - t0
        self.barrier()
        elapsed = time.perf_counter() - t0
        self.rank_ms = self.spread(own, steps)
        timing, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None
        per_kernel = {}
        for name, e0, e1, units, nbytes in timing:
            d = per_kernel.setdefault(name, {"s": 0.0, "units": 0, "launches": 0, "bytes": 0})
            d["s"] += e0.elapsed_time(e1) * 1e-3
            d["units"] += int(units)                      # (a device counter where the MLP ran on a compacted list)
            d["launches"] += 1
            d["bytes"] += int(nbytes or 0)
        return ana.max_over_ranks(elapsed, self.dev), per_kernel, out


def mlp_roofline(per_kernel, key, mode, flop_per_point, label):
    k = per_kernel.get(key, {"s": 0.0, "units": 0, "launches": 0})
    achieved = (k["units"] * flop_per_point / k["s"] / 1e12) if k["s"] > 0 else 0.0
    return {"kernel": label, "bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[mode], "unit": "TFLOP/s",
            "frac": achieved / PEAK_TFLOPS[mode], "traffic": None, "launches": k["launches"],
            "avg_launch_ms": (k["s"] / k["launches"] * 1e3) if k["launches"] else None, "flop_per_point": flop_per_point,
            "points": k["units"]}


def rescale_sigma(model, gain, mode, dev):
    """random-init sigma is ~0.017 +- 0.003 (one sign everywhere): every ray renders white or saturates on the far sample.
    Spread it about its median so that compositing, importance sampling and the PSNR mean something.  Same FLOPs, same
    kernels; only the numbers in the sigma row change."""
    if gain <= 0:
        return
    g = torch.Generator().manual_seed(5)
    probe = (torch.rand(1, 4096, 3, generator=g) * 1.6 - 0.8).to(dev)
    with torch.no_grad():
        for net in (model.nerf, model.nerf_fine):
            net.mlp_mode = "f32"
            med = net(probe)[1].median().item()
            net.mlp_mode = mode
            net.sigma.weight.mul_(gain)
            net.sigma.bias.mul_(gain).add_(-gain * med)


def render_bench(args, ctx, use_warp, mode, steps, warmup, dense=False, scaling="weak", checks=False, check_rays=None, one_pass=False):
    """BASELINE configs[1] / configs[2]: a full frame per step.
    one_pass (configs[1] only): the frame through the one-pass ray-march kernel (anr_ray_march) instead of the staged launches."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    dev, rank, world = ctx.dev, ctx.rank, ctx.world
    tbl = syn.make_smpl_table(0)
    torch.manual_seed(0)
    model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=use_warp, use_knn=True,
                         use_fine=True, mlp_mode=mode).eval().to(dev)
    rescale_sigma(model, args.sigma_gain, mode, dev)
    model.skip_invalid_samples = not dense
    vr = ana.VolumeRenderer(n_coarse=args.n_coarse, n_fine=args.n_fine, white_bkgd=True)
    one_pass = bool(one_pass or getattr(args, "one_pass", False)) and args.n_coarse == 64 and args.n_fine == 64
    vr.one_pass = one_pass
    H = W = args.hw
    c2w, focal, cen = syn.pinhole_camera(H, W)
    strong = scaling == "strong" and world > 1
    # weak: each rank renders its own frame (same camera, rank-seeded pose for the warp workload);
    # strong: every rank sets up the SAME frame and renders its contiguous slice of the rays
    pose_np = syn.animated_pose_params(seed=100 + (0 if strong else rank)) if use_warp else syn.static_pose_params()
    pose = {k: torch.from_numpy(v).to(dev) for k, v in pose_np.items()}
    templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), H, W, focal.tolist(), 0.1, 10.0, cen.tolist()).view(1, -1, 8)
    n_frame = rays.shape[1]
    if strong:
        lo, hi = ana.shard_range(n_frame, rank, world)
        rays = rays[:, lo:hi].contiguous()
    n_rays = rays.shape[1]

    def step():
        return ana.batched_inference(vr, model, rays, pose, templ, chunk=args.chunk)

    elapsed, per_kernel, out = ctx.timed(step, steps, warmup)
    total_rays = (n_frame if strong else n_frame * world) * steps
    evals = args.n_coarse + (args.n_coarse + args.n_fine if args.n_fine else 0)
    mlp_key = ("ray_march_warp" if use_warp else "ray_march") if one_pass else "mlp_forward"
    mlp = per_kernel.get(mlp_key, {"units": 0})
    result = {
        "value": total_rays / elapsed, "unit": "rays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "scaling": "strong" if strong else "weak", "dtype": mode,
        "config": {
            "workload": ("BASELINE configs[1]: 1024x1024 render, 64 coarse + 64 fine, random-init 8x256 MLP x2"
                         + (" (sigma rows rescaled about their median)" if args.sigma_gain > 0 else "")
                         + ", fixed SMPL pose, no LBS warp" if not use_warp else
                         "BASELINE configs[2]: configs[1] + inverse-LBS / exact 4-NN canonical warp (V=6890, animated pose)"),
            "rays_per_step_per_gpu": n_rays, "n_coarse": args.n_coarse, "n_fine": args.n_fine, "mlp_evals_per_ray": evals,
            "chunk_rays": args.chunk,
            "sharding": (f"one frame's rays sliced over {world} GPUs (per-frame setup replicated, no collective)" if strong
                         else f"{world} independent frames (ray-parallel, no collective): the same camera on every rank"
                              + (", rank-seeded body pose" if use_warp else "")),
            # warp on: samples farther than dis_threshold from the body are sigma = -1e5 / weight 0 whatever the MLP
            # says; the MLP runs on the others only (identical image; dense evaluates all of them like the reference)
            "mlp_on_valid_samples_only": bool(model.evaluate_valid_only),
            "mlp_points_per_step": mlp["units"] // max(steps, 1),
        },
        "roofline": mlp_roofline(per_kernel, mlp_key, mode, MLP_FLOP_PER_POINT,
                                 f"ray_march_kernel<{mode}{', warp' if use_warp else ''}> (ONE launch per frame: stratified samples, point generation, "
                                 + ("inverse-LBS / exact 4-NN warp of EVERY sample, " if use_warp else "") + "Fourier encoding, both "
                                 "networks, compositing, importance sampling + merge; the fraction is the WHOLE kernel's)" if one_pass else
                                 f"mlp_kernel<{mode}> (fused Fourier encoding + 11 GEMMs)"),
        "kernel_time_share": {k: round(v["s"] / elapsed, 4) for k, v in per_kernel.items()},
    }
    if one_pass:
        # the same frame through the staged launches: the one-pass kernel must return their bits
        vr.one_pass = False
        idx = torch.arange(0, n_rays, max(n_rays // 65536, 1), device=dev)[:65536]
        sub = rays[:, idx].contiguous()
        with torch.no_grad():
            st = ana.batched_inference(vr, model, sub, pose, templ, chunk=args.chunk)
            vr.one_pass = True
            op = ana.batched_inference(vr, model, sub, pose, templ, chunk=args.chunk)
        result["config"]["launches_per_frame"] = f"1 ({'anr_ray_march_warp' if use_warp else 'anr_ray_march'}) + the per-frame set-up"
        if use_warp:
            result["config"]["mlp_on_valid_samples_only"] = False     # the one-pass kernel evaluates every sample (the reference's way)
        result["equals_staged_launches_bit_for_bit"] = bool(all(torch.equal(st[k], op[k]) for k in st))
        result["equality_check_rays"] = int(idx.numel())
    # Everything around the MLP (sampling, point generation / warp, compaction, compositing, importance sampling) against
    # the HBM roofline.  `achieved` = the bytes those launches move BY DESIGN (every input read once, every output written
    # once: what each wrapper in ops.py declares) / their summed HIP-event time; SURVEY.md section 8(d)'s compulsory figure
    # (52 B per ray + 36 B per sample, which counts a 16-B canonical point per sample even where no kernel moves one) is
    # kept beside it.  With the warp on the binding resource is VALU issue (exact KNN), not HBM.
    others = {k: v for k, v in per_kernel.items() if k not in ("mlp_forward", "ray_march", "ray_march_warp")}
    other_s = sum(v["s"] for v in others.values())
    moved = sum(v["bytes"] for v in others.values())
    comp_bytes = n_rays * steps * (52 + 36 * evals)
    result["roofline_hbm_kernels"] = {
        "kernels": {k: {"ms_per_step": v["s"] / steps * 1e3, "GB/s": (v["bytes"] / v["s"] / 1e9) if v["s"] > 0 and v["bytes"] else None}
                    for k, v in sorted(others.items())},
        "bound": "hbm" if not use_warp else "valu (exact 4-NN)",
        "achieved": moved / other_s / 1e9 if other_s > 0 else None, "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "frac": moved / other_s / (PEAK_HBM_GBS * 1e9) if other_s > 0 else None,
        "bytes_per_ray_by_design": moved / max(n_rays * steps, 1),
        "survey_compulsory_bytes_per_ray": 52 + 36 * evals,
        "ms_per_step": other_s / steps * 1e3,
    }
    # (SURVEY 8(d)'s figure prices bytes some of which no kernel of this design moves — the no-warp path never materialises a
    # 16-B canonical point: where it exceeds what the launches move it is not a fraction of anything, and is left out)
    if other_s > 0 and comp_bytes <= moved:
        result["roofline_hbm_kernels"]["frac_on_survey_compulsory_bytes"] = comp_bytes / other_s / (PEAK_HBM_GBS * 1e9)
    if not use_warp and H == 1024 and args.n_coarse == 64 and args.n_fine == 64:
        # the same kernels by the memory-side counters (FETCH_SIZE x 2 + WRITE_SIZE of the round's PMC passes, per frame):
        # what the launches really moved, rays and partial sectors included, over this run's HIP-event time
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "mlp_hbm_traffic.json")))
        if files and other_s > 0:
            with open(files[-1]) as fh:
                t = json.load(fh)
            grp = t.get("hbm_kernels_cfg2")
            if grp:
                per_frame = sum(v["read_GB_per_frame"] + v["write_GB_per_frame"] for v in grp.values()) * 1e9
                result["roofline_hbm_kernels"].update({
                    "traffic": per_frame, "traffic_unit": f"HBM bytes per frame by PMC counters ({os.path.relpath(files[-1], ROOT)}, commit {t.get('commit')})",
                    "traffic_stale": traffic_is_stale(t),
                    "frac_by_counters": per_frame * (n_rays / n_frame) * steps / other_s / (PEAK_HBM_GBS * 1e9)})
    if checks and rank == 0 and world == 1:
        if not args.no_psnr and check_rays is None:
            # PSNR of this mode's image vs the fp32 parity path (pinned to the reference) on a centre crop
            c = min(256, H)
            lo = (H - c) // 2
            idx = (torch.arange(lo, lo + c)[:, None] * W + torch.arange(lo, lo + c)[None]).reshape(-1).to(dev)
            for net in (model.nerf, model.nerf_fine):
                net.mlp_mode = "f32"
            ref = ana.batched_inference(vr, model, rays[:, idx].contiguous(), pose, templ, chunk=1 << 16)
            for net in (model.nerf, model.nerf_fine):
                net.mlp_mode = mode
            result["psnr_vs_fp32_path_db"] = _psnr(out["rgbs_fine"][:, idx].cpu(), ref["rgbs_fine"].cpu())
        if args.cpu_rays > 0:
            result["cpu_baseline"], pick, ref = cpu_baseline(args, tbl, model, rays, pose_np, use_warp, check_rays)
            # the oracle's output as the checker: the same rays through the HIP path (fp32 parity mode and the benchmarked mode)
            sub = rays[:, pick.to(dev)].contiguous()
            key = "rgbs_fine" if args.n_fine else "rgbs"
            check = {"rays": int(pick.numel()),
                     "note": "rays outside 1e-4 in f32 mode are accounted for one by one in tests/test_gpu_parity.py::"
                             "test_every_out_of_tolerance_ray_is_accounted_for (the reference's `denom < eps` branch of the "
                             "inverse cdf moves a fine sample by up to a bin on the last ulp of the coarse weights)"}
            for m in dict.fromkeys(("f32", mode)):
                for net in (model.nerf, model.nerf_fine):
                    net.mlp_mode = m
                got = ana.batched_inference(vr, model, sub, pose, templ, chunk=1 << 16)[key].cpu()
                err = (got - ref[key]).abs().max(-1).values / ref[key].abs().max(-1).values.clamp_min(1e-3)
                check[m] = {"max_rel_err_rgb": err.max().item(), "rays_within_1e-4": (err <= 1e-4).float().mean().item(),
                            "psnr_db": _psnr(got, ref[key])}
                if m == "f32":
                    # every ray outside 1e-4 re-rendered by the oracle with the HIP path's sampling decisions injected
                    # (tests/accounting.py: the checker, never on a timed path)
                    check[m]["accounting"] = _account(lambda acc: acc.account_for_rays(
                        model, vr, _oracle_table(tbl), sub.cpu(), {k: v.cpu() for k, v in pose.items()},
                        {k: v.cpu() for k, v in templ.items()}, ref, z_fine_ref=ref.get("_z_fine"), label="bench", quiet=True))
            for net in (model.nerf, model.nerf_fine):
                net.mlp_mode = mode
            result["oracle_check"] = check
    return result


def _oracle_table(tbl):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_table
    return oracle_table(tbl)


def _account(fn):
    """Run one of tests/accounting.py's checkers for an `oracle_check` entry: its statistics, or what it could not account for."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import accounting
    try:
        stats = fn(accounting)
        stats["every_difference_accounted_for"] = True
        return stats
    except AssertionError as exc:
        return {"every_difference_accounted_for": False, "unaccounted": str(exc)[:300]}


def grid_bench(args, ctx, mode, steps, warmup, dense=False):
    """BASELINE configs[4]: relu(sigma) of the fine field on a 512^3 grid around the posed body (extract_mesh.py:152-158),
    voxel-sharded: rank r evaluates slab r of the SAME grid (strong scaling), no collective on the data path."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    dev, rank, world = ctx.dev, ctx.rank, ctx.world
    N = 512
    tbl = syn.make_smpl_table(0)
    torch.manual_seed(0)
    model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True,
                         use_fine=True, mlp_mode=mode).eval().to(dev)
    rescale_sigma(model, args.sigma_gain or 1.0, mode, dev)
    with torch.no_grad():
        model.skip_invalid_samples = not dense
        pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=100).items()}
        templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
        rays = torch.zeros(1, 1, 8, device=dev)
        rays[..., 5], rays[..., 7] = -1, 10
        model.set_body_model(pose, templ)
        model.convert_to_body_model_space(rays)
        model.clac_ober2cano_transform()

        def step():
            return ana.sigma_grid(model, N, chunk=1 << 27, rank=rank, world=world)
        elapsed, per_kernel, (sig, _) = ctx.timed(step, steps, warmup)
        # what extract_mesh.py does with the grid next (:159-165), outside the timed region and on rank 0's slab only when the
        # grid is sharded: the level set by marching cubes (csrc/mesh.hip)
        mesh_ms = mesh_tris = None
        if world == 1:
            field = (5.0 - sig.view(N, N, N)).contiguous()
            ana.mesh.marching_cubes(field, 0.0)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            _, tris = ana.mesh.marching_cubes(field, 0.0)
            torch.cuda.synchronize(dev)
            mesh_ms, mesh_tris = (time.perf_counter() - t0) * 1e3, int(tris.shape[0])
        check = None
        if world == 1 and args.cpu_rays > 0:
            # the oracle on sampled voxels of THIS grid (half of them occupied ones): extract_mesh.py:27-61 restated on the CPU,
            # against the timed mode and — voxel by voxel, tests/accounting.py — against the fp32 parity mode
            import numpy as np
            g = torch.Generator().manual_seed(3)
            flat_occ = torch.nonzero(sig > 0)[:, 0]
            sel = torch.cat([flat_occ[torch.randint(0, flat_occ.numel(), (2048,), generator=g).to(dev)],
                             torch.randint(0, N ** 3, (2048,), generator=g).to(dev)])
            lin = np.linspace(-1.2, 1.2, N)
            a, b, c = (sel // (N * N)).cpu().numpy(), ((sel // N) % N).cpu().numpy(), (sel % N).cpu().numpy()
            center = (model.verts.max(dim=1)[0] + model.verts.min(dim=1)[0]) / 2.
            points = torch.from_numpy(np.stack([lin[b], lin[a], lin[c]], -1)).unsqueeze(0).float().to(dev) + center
            from oracle import animnerf_oracle as orc
            otbl = _oracle_table(tbl)
            st = dict(verts=model.verts.cpu(), ober2cano=model.ober2cano_transform.cpu())
            P = {k: v.detach().cpu() for k, v in model._net(True).named_parameters()}
            want = torch.relu(orc.field_query(P, points.cpu(), st, otbl["lbs_weights"], True, model.dis_threshold, chunk=1024)[1][0, :, 0])
            got = sig[sel].cpu()
            scale = want[want > 0].median().item() if (want > 0).any() else 1.0
            check = {"voxels": int(sel.numel()), "occupied_in_oracle": int((want > 0).sum()),
                     mode: {"occupancy_agreement": ((got > 0) == (want > 0)).float().mean().item(),
                            "max_abs_err_over_median_sigma": ((got - want).abs().max() / scale).item(),
                            "voxels_within_1e-4": ((got - want).abs() <= 1e-4 * want.abs() + 1e-4 * scale).float().mean().item()}}
            for net in (model.nerf, model.nerf_fine):
                net.mlp_mode = "f32"
            sig32 = ana.sigma_grid(model, N, chunk=1 << 27)[0]
            check["f32"] = {"voxels_within_1e-4": ((sig32[sel].cpu() - want).abs() <= 1e-4 * want.abs() + 1e-4 * scale).float().mean().item(),
                            "accounting": _account(lambda acc: acc.account_for_points(model, otbl, points, sig32[sel], use_fine=True, relu=True,
                                                                                      dis_threshold=model.dis_threshold, label="bench cfg5", quiet=True))}
            if mode != "f32":
                # the timed mode's LEVEL SET against the parity mode's, as what extract_mesh.py:159-165 makes of the two grids:
                # vertex-to-surface distances both ways (tests/accounting.py; gated at >= 99.7 % within one voxel by
                # tests/test_gpu_parity.py::test_timed_mode_mesh_within_one_voxel_of_the_parity_mode_mesh)
                def _mesh_gap(acc):
                    va, ta = ana.mesh.marching_cubes((5.0 - sig.view(N, N, N)).contiguous(), 0.0)
                    vb, tb = ana.mesh.marching_cubes((5.0 - sig32.view(N, N, N)).contiguous(), 0.0)
                    out = {"occupancy_flips": int(((sig > 5.0) != (sig32 > 5.0)).sum()), "occupied_f32": int((sig32 > 5.0).sum())}
                    for name, (p_, q_, t_) in ((mode + "_to_f32", (va, vb, tb)), ("f32_to_" + mode, (vb, va, ta))):
                        dd = acc.vertex_to_surface_distance(p_, q_, t_, N, reach=2)
                        out[name] = {"vertices": int(dd.numel()), "within_1_voxel": round(float((dd <= 1.0).float().mean()), 5),
                                     "within_2_voxels": round(float((dd <= 2.0).float().mean()), 5), "mean_voxels": round(float(dd.mean()), 4)}
                    return out
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import accounting as _acc
                check["mesh_vs_f32_mesh"] = _mesh_gap(_acc)
            for net in (model.nerf, model.nerf_fine):
                net.mlp_mode = mode
    return {
        "metric": "grid points/sec, 512^3 sigma query (mesh extraction input)", "value": N ** 3 * steps / elapsed,
        "unit": "points/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "scaling": "strong", "dtype": mode,
        "config": {"workload": "BASELINE configs[4]: 512^3 sigma grid around the posed body, fine network, voxel-sharded "
                               "(contiguous slabs, no collective)", "grid": N, "mlp_on_valid_voxels_only": bool(model.evaluate_valid_only),
                   "mlp_points_per_step_this_rank": per_kernel.get("mlp_forward", {"units": 0})["units"] // max(steps, 1),
                   "occupied_voxels_this_rank": int((sig > 0).sum()),
                   "marching_cubes_after_the_timed_region": {"ms": mesh_ms, "triangles": mesh_tris, "threshold": 5.0}},
        "roofline": mlp_roofline(per_kernel, "mlp_forward", mode, MLP_FLOP_SIGMA_ONLY, f"mlp_kernel<{mode}, sigma only>"),
        **({"oracle_check": check} if check else {}),
    }


def train_bench(args, ctx, mode, steps, warmup, dense=False, scaling="weak", frames=None, refine=False):
    """BASELINE configs[3] shape: one optimisation step = `frames_per_gpu` frames x 32x32 rays, 64 coarse + 32 fine,
    perturb = 1, rgb/alpha/foreground/background/normals losses, backward, gradient all-reduce (RCCL), Adam.
    scaling="strong": the reference's own partitioning (train.py:81-86,451-458, config.py:77 `strategy='dp'`,
    configs/people_snapshot/male-3-casual.yaml `batch_size: 16`): ONE batch of 16 frames split over the ranks, 16 // world
    frames each; every rank's loss is the mean over ITS frames and the gradients are averaged over the ranks — DataParallel's
    mean over replicas of per-replica means.  scaling="weak": `frames_per_gpu` frames on every rank (global batch grows)."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    dev, rank, world = ctx.dev, ctx.rank, ctx.world
    tbl = syn.make_smpl_table(0)
    torch.manual_seed(0)
    model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True,
                         use_fine=True, mlp_mode=mode).to(dev)
    model.skip_invalid_samples = not dense
    refine = refine or getattr(args, "refine", False)
    if refine:
        # configs/people_snapshot/male-3-casual_refine.yaml:52-53 + train.py:433-437: the trained networks are loaded and
        # FROZEN (pretrained_model_requires_grad: False); optim_body_params stays on — only the pose rows train
        for p in model.parameters():
            p.requires_grad_(False)
    hp = ana.TrainHParams(n_samples=64, n_importance=32, chunk=2048)
    GLOBAL = 16                                                # batch_size of the shipped config
    strong = scaling == "strong"
    if strong and GLOBAL % world:
        raise ValueError(f"cfg4 --scaling strong splits {GLOBAL} frames: the rank count must divide it")
    F = GLOBAL // world if strong else (frames or args.frames_per_gpu)
    table = ana.BodyModelParams(114).to(dev)                  # 114 training frames (configs/people_snapshot/male-3-casual.yaml)
    seeded = syn.animated_pose_params(seed=200, bs=114)        # the SAME table on every rank: its gradients are all-reduced
    for name in table.param_names:                            # optim_body_params: True
        table.init_parameters(name, torch.from_numpy(seeded[name]).to(dev), requires_grad=True)
    # the step replayed from a HIP graph (Trainer.step_graphed; ~190 launches per step otherwise, and the host, not the GPU,
    # sets the pace on a slow box): one rank — the whole step; more ranks — forward + backward, then the all-reduce of the two
    # flat gradient buffers and Adam.  ANR_BENCH_NO_GRAPH=1: eager, the bucketed all-reduce overlapping backward.
    graphed = dev.type == "cuda" and not os.environ.get("ANR_BENCH_NO_GRAPH")
    trainer = ana.Trainer(model, ana.VolumeRenderer(n_coarse=64, n_fine=32), hp, body_model_params=table, graph=graphed)
    trainer.renderer.reuse_coarse_warp = not os.environ.get("ANR_BENCH_NO_WARP_REUSE")     # (A/B switch for the fine pass's copy)
    if strong:                                                # this rank's slice of the ONE global batch (DP's scatter on dim 0)
        frame_idx = (torch.arange(GLOBAL, device=dev) * (114 // GLOBAL))[rank * F:(rank + 1) * F]
    else:
        frame_idx = (torch.arange(F, device=dev) * (114 // F) + rank) % 114  # a rank's own frames, as a distributed sampler deals them
    c2w, focal, cen = syn.pinhole_camera(32, 32)
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 32, 32, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(F, 1, 1, 1)
    templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
    g = torch.Generator().manual_seed(0 if strong else rank)
    nb = GLOBAL if strong else F                               # strong: targets of the global batch, sliced like the frames
    cut = slice(rank * F, (rank + 1) * F) if strong else slice(None)
    rgbs = torch.rand(nb, 32, 32, 3, generator=g)[cut].to(dev)
    alphas = (torch.rand(nb, 32, 32, 1, generator=g) > 0.5).float()[cut].to(dev)
    fg = (torch.rand(nb, 128, 3, generator=g) * 0.4 - 0.2)[cut].to(dev)
    bg = (torch.rand(nb, 128, 3, generator=g) * 2 - 1)[cut].to(dev)
    last = {}

    def step():
        last["loss"], _ = trainer.step_graphed(rays, rgbs, alphas, None, templ, fg, bg, perturb=1.0, frame_idx=frame_idx)
        return last["loss"]

    def eager_step():
        last["loss"], _ = trainer.step(rays, rgbs, alphas, None, templ, fg, bg, perturb=1.0, frame_idx=frame_idx)
        return last["loss"]

    if dev.type == "cuda":
        torch.cuda.reset_peak_memory_stats(dev)
    with trainer.loop():                # (the Trainer's own stream for the whole loop: no stream fences per step)
        if graphed:
            # untimed set-up, like a compilation: GRAPH_WARM_STEPS eager steps, then the capture (which executes nothing)
            while trainer._graph is None:
                step()
        elapsed, per_kernel, loss = ctx.timed(step, steps, warmup)
        eager_elapsed, eager_steps = elapsed, steps
        if graphed:
            # a replay launches nothing from Python, so the per-kernel HIP events come from a few EAGER steps of the same batch
            # right after the timed region (same kernels, same shapes; their host-side pace does not enter any number below)
            # ... with every launch on ONE stream: the replayed step runs its normals branch and its weight gradients next to the
            # render passes (fused_step.py), and a kernel's time next to another kernel says nothing about the kernel
            eager_steps = min(steps, 5)
            branches = getattr(trainer.explicit, "parallel", None)
            if branches:
                trainer.explicit.parallel = False
            eager_elapsed, per_kernel, _ = ctx.timed(eager_step, eager_steps, 0)
            if branches:
                trainer.explicit.parallel = True
    n_rays = F * 1024
    return {
        "metric": "rays/sec, training step (64+32 samples, 256-wide MLP x2, fwd+bwd+Adam)",
        "value": n_rays * steps * world / elapsed, "unit": "rays/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "scaling": scaling, "dtype": mode,
        # (activation buffers are sized for "every sample valid": the row counts stay on the device — INTEGRATION.md)
        "peak_alloc_GiB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2) if dev.type == "cuda" else None,
        "config": {"global_batch_frames": F * world, "frames_per_gpu": F,
                   "partitioning": ("ONE batch of 16 frames split over the ranks (the reference's DP scatter on dim 0), gradients = mean "
                                    "over ranks of per-rank means" if strong else "frames_per_gpu frames on every rank: the global batch grows with N"),
                   "workload": "BASELINE configs[3] shape: train step, %d frames x 32x32 rays per GPU, 64 coarse + 32 fine, "
                               "perturb=1, rgb + alpha + fg/bg + normals losses (reference defaults), pose refinement on (optim_body_params), "
                               "%s + Adam" % (F, "ONE gradient all-reduce (4.8 MB flat buffer, averaged by the collective) after the replayed backward" if graphed
                                              else "bucketed gradient all-reduce (2 x 2.4 MB, overlapped with backward)")
                               + ("; the `_refine` stage: both networks FROZEN (pretrained_model_requires_grad False), only the pose "
                                  "rows train — no weight-gradient launch, sign bits instead of saved activations" if refine else ""),
                   "networks_frozen": bool(refine),
                   "mlp_on_valid_samples_only": bool(model.evaluate_valid_only),
                   "mlp_rows_per_step": per_kernel.get("mlp_forward_bits" if refine else "mlp_forward_save", {"units": 0})["units"] // max(eager_steps, 1),
                   "rays_per_step_per_gpu": n_rays, "grad_floats": sum(p.numel() for p in trainer.params),
                   "kernel_launches_timed_per_step": sum(v["launches"] for v in per_kernel.values()) // max(eager_steps, 1),
                   "step_launch": (("one HIP graph replay per step" if world == 1 else "forward + backward as one HIP graph replay per step")
                                   + " (captured after %d eager steps, untimed; three parallel branches); per-kernel times from "
                                   "%d eager single-stream steps after the timed region, %.2f ms per step there"
                                   % (trainer.GRAPH_WARM_STEPS, eager_steps, eager_elapsed / eager_steps * 1e3)) if graphed
                   else "eager: one launch per kernel"},
        "roofline": mlp_roofline(per_kernel, "mlp_forward_bits" if refine else "mlp_forward_save", mode, MLP_FLOP_PER_POINT,
                                 f"mlp_kernel<{mode}, {'sign bits only' if refine else 'save'}> (training forward)"),
        "kernel_time_share": {k: round(v["s"] / eager_steps / (elapsed / steps), 4) for k, v in per_kernel.items()},
        "final_loss": float(loss),
    }


def train_bench_child(args, ctx, mode, steps, warmup, frames=None, refine=False):
    """cfg4 as an extra of the default run: the graphed training step in a CHILD process (`bench.py --workload cfg4
    --no-extras`), its line merged into this one.  A HIP-graph replay that goes wrong takes its process down with a GPU memory
    fault (tools/soak_train.py found one such sequence: > 50 replays, then a device synchronisation followed by scalar reads,
    then more replays — not what this workload does, but a fault here must not cost the headline line).  If the child does
    not deliver, the step is measured eagerly in this process and the entry says so."""
    import subprocess
    if ctx.world != 1 or ctx.dev.type != "cuda" or os.environ.get("ANR_BENCH_NO_GRAPH"):
        return train_bench(args, ctx, mode, steps, warmup, frames=frames, refine=refine)
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "cfg4", "--no-extras", "--steps", str(steps), "--warmup", str(warmup),
           "--mode", mode, "--frames-per-gpu", str(frames or args.frames_per_gpu)] + (["--refine"] if refine else [])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    note = None
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            got = json.loads(lines[-1])
            got["config"]["process"] = "child process of the default run (graph replays isolated from the headline's process)"
            return got
        note = f"graphed child exited with {r.returncode}: {r.stderr[-200:]!r}"
    except Exception as exc:                                    # noqa: BLE001
        note = f"graphed child failed: {type(exc).__name__}: {exc}"[:300]
    os.environ["ANR_BENCH_NO_GRAPH"] = "1"
    try:
        got = train_bench(args, ctx, mode, steps, warmup, frames=frames, refine=refine)
    finally:
        os.environ.pop("ANR_BENCH_NO_GRAPH", None)
    got["config"]["process"] = "eager step in the parent: " + note
    return got


TRAFFIC_SOURCES = ("anim-nerf_amd/csrc/mlp_core.h", "anim-nerf_amd/csrc/mlp.hip", "anim-nerf_amd/csrc/mlp_inst_bf16.hip",
                   "anim-nerf_amd/csrc/composite.hip", "anim-nerf_amd/csrc/composite_core.h")


def kernel_sources_sha():
    """sha256 over the sources of the kernels whose PMC traffic profiles/rNN/mlp_hbm_traffic.json records (tools/traffic_json.py
    stores it at collection time; there is no .git on the GPU box to ask for ancestry)"""
    import hashlib
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic_is_stale(t):
    return t.get("kernel_sources_sha") != kernel_sources_sha()


def measured_traffic(mode, variant, points_per_launch):
    """HBM bytes per launch of the MLP kernel: bytes per point from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs, gfx950 correction applied) x points per launch.  The newest profiles/rNN/mlp_hbm_traffic.json wins; the
    file names the commit it was measured at."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "mlp_hbm_traffic.json")))
    if mode != "bf16" or not files or not points_per_launch:
        return None, None
    with open(files[-1]) as fh:
        t = json.load(fh)
    key = {"indexed": "bytes_per_point_indexed", "explicit": "bytes_per_point_explicit", "rays": "bytes_per_point"}[variant]
    if key not in t:
        return None, None
    return t[key] * points_per_launch, (f"HBM bytes per launch = {t[key]:.1f} B per point (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                        f"separate passes; {os.path.relpath(files[-1], ROOT)}, measured at commit "
                                        f"{t.get('commit', 'of round 1')}) x points per launch"
                                        + ("; STALE: the kernel sources have changed since" if traffic_is_stale(t) else
                                           "; the kernel sources are the ones it was measured on"))


def collective_fields(ctx):
    """what the line says about the process group: ranks counted by a real all-reduce on the data backend, and the spread of
    the ranks' own clocks over the headline's timed region"""
    return {"collective_backend": (ctx.backend if ctx.world > 1 else None),
            "rccl_ranks": ctx.collective_ranks if ctx.backend == "nccl" and ctx.world > 1 else None,
            "collective_ranks": ctx.collective_ranks,
            "rank_ms_per_step": {"min": ctx.rank_ms[0], "max": ctx.rank_ms[1]} if ctx.rank_ms else None}


def plumbing_only(args, ctx):
    """--plumbing-only: everything bench.py does around the kernels at N ranks, with a sleep as the step"""
    def step():
        time.sleep(0.01 * (1 + ctx.rank))                 # rank r is slower than rank r - 1
        return None
    elapsed, _, _ = ctx.timed(step, args.steps, args.warmup)
    ok = ctx.all_ok(True)
    result = {"metric": "plumbing only (no kernels)", "value": ctx.world * args.steps / elapsed, "unit": "steps/s",
              "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": None, "data": "none",
              "config": {"workload": "sleep(10 ms x (rank + 1))"}, "all_ranks_ok": ok, **collective_fields(ctx)}
    if ctx.rank == 0:
        print(json.dumps(result), flush=True)
    if ctx.world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                             # never returns
    if args.extras_only:
        return extras_only(args)                      # never returns
    ctx = Ctx(args)
    if args.plumbing_only:
        return plumbing_only(args, ctx)
    import anim_nerf_amd as ana
    ana._lib.load()                                   # fails loudly if the HIP library is missing
    rank, world = ctx.rank, ctx.world

    if args.workload == "cfg4":
        result = train_bench(args, ctx, args.mode, args.steps, args.warmup, args.dense, scaling=args.scaling)
    elif args.workload == "cfg5":
        result = grid_bench(args, ctx, args.mode, args.steps, args.warmup, args.dense)
    else:
        use_warp = args.workload == "cfg3"
        result = render_bench(args, ctx, use_warp, args.mode, args.steps, args.warmup, args.dense, args.scaling, checks=True)
        result = {"metric": "rays/sec (64+64 samples, 256-wide MLP)", **result}
        r = result["roofline"]
        variant = "rays" if not use_warp else ("indexed" if result["config"]["mlp_on_valid_samples_only"] else "explicit")
        r["traffic"], r["traffic_unit"] = measured_traffic(args.mode, variant, r["points"] / r["launches"] if r["launches"] else 0)
        r["traffic_stale"] = None if r["traffic"] is None else ("STALE" in r["traffic_unit"])
    result.update({"higher_is_better": True, "vs_baseline": None, "data": "synthetic", **collective_fields(ctx)})

    extras = not args.no_extras and args.workload == "cfg2" and args.scaling == "weak"
    if extras and world == 1:
        result.update(collect_extras(args, ctx))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    if extras and world > 1:
        # N > 1: every other line this repository quotes is measured by a JOB OF ITS OWN — a fresh `torch.distributed.run` child
        # with the same rank count, started by rank 0 once the headline's process group is gone (a child process, never an exec
        # of this one).  Whatever happens in there — a GPU fault in a replayed training graph, a hang in a collective, an OOM on
        # one GPU — the headline above is already measured and is printed below, once, with the child's outcome beside it.
        if rank != 0:
            return
        del ctx
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        result.update(extras_child_job(args, world))
    if rank == 0:
        print(json.dumps(result), flush=True)
    if result.get("extras_failed") and (os.environ.get("ANR_BENCH_STRICT") or (world > 1 and not os.environ.get("ANR_BENCH_LENIENT"))):
        sys.exit(3)                                    # (the line above is complete and valid; the exit code says an extra is missing)


def collect_extras(args, ctx):
    """Every other line this repository quotes, each with its own warm-up and timed region: {"modes", "workloads"[, "extras_failed"]}.
    One rank: in this process (the graphed training steps in children of their own, train_bench_child).  More ranks: this IS the
    child job (`--extras-only`, extras_child_job); after each workload the ranks agree on its outcome (one MIN all-reduce, like
    every collective here with a timeout) and the first failure on any rank ends the job with what has been measured."""
    rank, world = ctx.rank, ctx.world
    out, w = {}, {}

    def brief(r, *keep):
        keys = ("value", "unit", "ms_per_step", "steps", "warmup", "n_gpus", "scaling", "dtype", "roofline", "peak_alloc_GiB") + keep
        return {**{k: r[k] for k in keys if k in r}, "config": r["config"]}

    class Stop(Exception):
        pass

    def extra(fn, *a, keep=(), **kw):
        try:
            # every workload starts with an empty allocator cache: the previous one's blocks (a training graph's private
            # pool, 2-GB grid chunks) otherwise decide whether this one's first allocations are cache hits or hipMallocs
            import gc
            gc.collect()
            if ctx.dev.type == "cuda":
                torch.cuda.empty_cache()
            got, err = brief(fn(*a, **kw), *keep), None
        except Exception as exc:                        # noqa: BLE001
            got, err = None, f"{type(exc).__name__}: {exc}"[:300]
        if world > 1 and not ctx.all_ok(err is None):
            # a failure need not be symmetric (OOM or a HIP error on ONE GPU): every rank knows now, and nobody walks into the
            # next workload's barrier
            raise Stop(err or "failed on another rank")
        return got if err is None else {"error": err}
    name = None
    try:
        if world == 1:
            out["modes"] = {"f32": extra(render_bench, args, ctx, False, "f32", 2, 1, keep=("roofline_hbm_kernels",))}
            # the headline workload through the one-pass ray-march kernel (csrc/ray_march.hip): one launch per frame, same bits
            out["modes"]["one_pass"] = extra(render_bench, args, ctx, False, args.mode, 4, 1, one_pass=True,
                                             keep=("equals_staged_launches_bit_for_bit", "equality_check_rays", "kernel_time_share"))
            # (with its own oracle check: 1,024 rays through the oracle's brute-force 4-NN warp, ~25 s of host time)
            w["cfg3"] = extra(render_bench, args, ctx, True, args.mode, 3, 1, checks=True, check_rays=1024,
                              keep=("roofline_hbm_kernels", "kernel_time_share", "oracle_check", "cpu_baseline"))
            w["cfg3_dense"] = extra(render_bench, args, ctx, True, args.mode, 2, 1, dense=True)
            # configs[2] as ONE launch per frame: the warp inside the one-pass kernel, every sample through the networks (the dense
            # evaluation, like cfg3_dense and the reference) — the literal single pass of the north star next to the staged default
            w["cfg3_one_pass"] = extra(render_bench, args, ctx, True, args.mode, 2, 1, one_pass=True,
                                       keep=("equals_staged_launches_bit_for_bit", "equality_check_rays", "kernel_time_share"))
            w["cfg4"] = extra(train_bench_child, args, ctx, args.mode, 8, 4, keep=("kernel_time_share", "final_loss"))
            # the per-rank step of configs[3] on 8 GPUs: 2 of the batch's 16 frames (strong scaling's unit of work, on one GPU)
            w["cfg4_f2"] = extra(train_bench_child, args, ctx, args.mode, 16, 4, frames=2, keep=("kernel_time_share", "final_loss"))
            # the `_refine` stage of the shipped configs (networks loaded and frozen, the poses train: train.py:433-437)
            w["cfg4_refine"] = extra(train_bench_child, args, ctx, args.mode, 8, 4, refine=True, keep=("kernel_time_share", "final_loss"))
            w["cfg5"] = extra(grid_bench, args, ctx, args.mode, 3, 2, keep=("oracle_check",))
        else:
            # N > 1: the strong-scaling counterpart of the headline (one frame's rays sliced over the ranks), the warp
            # workload, and the training step with its RCCL gradient all-reduce
            for name, fn, a, kw in (
                    ("cfg2_strong", render_bench, (False, args.mode, 3, 1), dict(scaling="strong")),
                    ("cfg3", render_bench, (True, args.mode, 3, 1), {}),
                    ("cfg3_strong", render_bench, (True, args.mode, 3, 1), dict(scaling="strong")),
                    ("cfg5", grid_bench, (args.mode, 3, 2), {}),
                    ("cfg4", train_bench, (args.mode, 8, 4), dict(keep=("final_loss",))),
                    # configs[3] as the reference shards it: 16 frames / N per rank
                    ("cfg4_strong", train_bench, (args.mode, 16, 4), dict(scaling="strong", keep=("final_loss",)))):
                if name == "cfg4_strong" and 16 % world:
                    continue
                w[name] = extra(fn, args, ctx, *a, **kw)
    except Stop as stop:
        w[name or "extras"] = {"error": str(stop)}
    out["workloads"] = w
    failed = [k for k, v in {**w, **out.get("modes", {})}.items() if isinstance(v, dict) and "error" in v]
    if failed:
        out["extras_failed"] = True
    return out


def extras_only(args):
    """`--extras-only` (the child job of an N > 1 run): the process group of its own, collect_extras, rank 0 prints the object."""
    ctx = Ctx(args)
    import anim_nerf_amd as ana
    ana._lib.load()
    out = collect_extras(args, ctx)
    if ctx.rank == 0:
        print(json.dumps({"metric": "extras", **out}), flush=True)
    sys.stdout.flush()
    if ctx.world > 1:
        # best-effort teardown (after a failure a rank may never answer: bounded), then leave
        import threading
        import torch.distributed as dist
        t = threading.Thread(target=lambda: dist.destroy_process_group(), daemon=True)
        t.start()
        t.join(20.0)
    os._exit(3 if out.get("extras_failed") else 0)


def extras_child_job(args, world):
    """Rank 0 of an N > 1 run, after its own process group is gone: run `bench.py --gpus N --extras-only` as a child
    `torch.distributed.run` job and hand back its object — or the reason it did not deliver."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(world), "--extras-only", "--mode", args.mode,
           "--hw", str(args.hw), "--n-coarse", str(args.n_coarse), "--n-fine", str(args.n_fine), "--chunk", str(args.chunk),
           "--sigma-gain", str(args.sigma_gain), "--frames-per-gpu", str(args.frames_per_gpu), "--cpu-rays", "0", "--no-psnr"]
    scrub = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
             "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE")
    env = {k: v for k, v in os.environ.items() if k not in scrub and not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_ASYNC"))}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    limit = float(os.environ.get("ANR_BENCH_EXTRAS_TIMEOUT", "1500"))
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric": "extras"' in ln]
        if lines:
            got = json.loads(lines[-1])
            got.pop("metric", None)
            got["extras_process"] = (f"a child torch.distributed.run job of {world} ranks started after the headline's process group was "
                                     f"destroyed: exit code {r.returncode}, {time.perf_counter() - t0:.0f} s")
            if r.returncode != 0:
                got["extras_failed"] = True
            return got
        note = f"extras job exited with {r.returncode} and no line: {r.stderr[-300:]!r}"
    except Exception as exc:                                    # noqa: BLE001
        note = f"extras job failed: {type(exc).__name__}: {exc}"[:400]
    return {"workloads": {"error": note}, "extras_failed": True}


def cpu_baseline(args, tbl, model, rays, pose_np, use_warp, max_rays=None):
    """The oracle (CPU restatement of the reference, torch CPU ops, all host cores) on a bounded
    sample of the same workload: `cpu_rays` rays of the same frame, same sample counts."""
    from anim_nerf_amd import synthetic as syn
    from oracle import animnerf_oracle as orc
    import anim_nerf_amd as ana
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(cores)
    bm = ana.SMPL(data_struct=tbl)
    otbl = dict(v_template=bm.v_template, shapedirs=bm.shapedirs, posedirs=bm.posedirs.T.contiguous().T,      # (the reference's layout)
                J_regressor=bm.J_regressor,
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
    n = int(min(max_rays or args.cpu_rays, args.cpu_rays, max(64, rate * args.cpu_seconds)))
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
