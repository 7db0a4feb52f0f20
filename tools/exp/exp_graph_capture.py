#!/usr/bin/env python3
"""Which pieces of the training step survive HIP graph capture + instantiate + replay?  One stage per process:
    python tools/exp/exp_graph_capture.py <stage>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import anim_nerf_amd as ana                                  # noqa: E402
from anim_nerf_amd import synthetic as syn                   # noqa: E402

full_stage = sys.argv[1]
stage = full_stage.replace("_bump", "") if not full_stage.startswith("trainer") else full_stage
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
torch.manual_seed(0)
m = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True, use_fine=True, mlp_mode="bf16").to(dev)
F, H = 4, 16
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=200, bs=F).items()}
c2w, focal, cen = syn.pinhole_camera(H, H)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), H, H, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(F, 1, 1, 1)
vr = ana.VolumeRenderer(n_coarse=32, n_fine=16)
pts = torch.rand(4096, 4, device=dev)
pts[:, 3] = 1.0


_g = torch.Generator().manual_seed(17)
loss_rgb = torch.rand(F, H, H, 3, generator=_g).to(dev)
loss_alp = (torch.rand(F, H, H, 1, generator=_g) > 0.5).float().to(dev)


def body():
    if stage == "torch":
        return (pts * 2).sum()
    if stage == "raygen":
        return ana.gen_rays(torch.from_numpy(c2w).to(dev), H, H, focal.to(dev) if torch.is_tensor(focal) else focal.tolist(), 0.1, 10.0, cen.tolist())
    if stage == "pack":
        m.nerf._pack_cache.clear()
        pack, mode = m.nerf.weight_pack("bf16")
        return ana.ops.mlp_forward(pack, mode, pts)
    if stage == "packbwd":
        sd = {k: v for k, v in m.nerf.state_dict().items()}
        return ana.ops.mlp_pack(sd, 2, backward=True).float()
    if stage == "mlp":
        pack, mode = m.nerf.weight_pack("bf16")
        return ana.ops.mlp_forward(pack, mode, pts)
    if stage == "smpl":
        with torch.no_grad():
            m.set_body_model(pose, templ)
            return m.verts
    if stage in ("sysfwd_simple", "sysfwd_loss", "sysfwd_termloss"):
        # system_forward + (simple | fused | term-by-term) loss + backward
        res = ana.system_forward(vr, m, rays, pose, templ, perturb=0.0, chunk=1 << 20)
        if stage == "sysfwd_simple":
            loss = res["rgbs_fine"].square().mean() + res["rgbs"].square().mean()
        else:
            hp_l = ana.TrainHParams(n_samples=32, n_importance=16, lambda_normals=0.0, fused_losses=stage == "sysfwd_loss")
            loss = ana.compute_loss(m, hp_l, loss_rgb, loss_alp, res, None, None)[0]
        loss.backward()
        return loss.detach()
    if stage in ("warp", "render", "render_grad", "step", "fwd_bwd"):
        ctx = torch.no_grad() if stage in ("warp", "render") else torch.enable_grad()
        with ctx:
            m.set_body_model(pose, templ)
            rb = m.convert_to_body_model_space(rays.view(F, -1, 8))
            m.clac_ober2cano_transform()
            if stage == "warp":
                z = vr.sample_coarse(rb)
                return ana.ops.warp_points(m.knn_index(), m.ober2cano_transform, m.body_model.lbs_weights, 0.2, rays=rb, z=z,
                                           skip_far=True, lean=True)[0]
            out = vr(m, rb, perturb=0.0)
            if stage == "render":
                return out["rgbs_fine"]
            loss = out["rgbs_fine"].square().mean() + out["rgbs"].square().mean()
            if stage == "render_grad":
                return loss
            loss.backward()
            return loss.detach()
    if stage.startswith("trainer"):
        return trainer_body()
    raise SystemExit("unknown stage")


if stage.startswith("trainer"):
    table = ana.BodyModelParams(40).to(dev)
    seeded = syn.animated_pose_params(seed=200, bs=40)
    for name in table.param_names:
        table.init_parameters(name, torch.from_numpy(seeded[name]).to(dev), requires_grad=True)
    hp = ana.TrainHParams(n_samples=32, n_importance=16, lambda_normals=0.0 if "nonormals" in stage else 0.01)
    tr = ana.Trainer(m, vr, hp, body_model_params=table if "table" in stage or stage == "trainer" else None, graph=True)
    if "adamA" in stage or "adamB" in stage or "adamC" in stage:
        groups = [{"params": g["params"], "lr": (float(g["lr"]) if "adamA" in stage else g["lr"])} for g in tr.optimizer.param_groups]
        tr.optimizer = torch.optim.Adam(groups, eps=1e-8, weight_decay=0, capturable=True,
                                        **({"foreach": True} if "adamB" in stage else {"fused": True}))
    if "adamD" in stage:
        import types
        opt = tr.optimizer
        def only_step():
            for p in tr.params:
                if p.grad is not None:
                    p.data.add_(p.grad, alpha=-1e-3)
        tr.optimizer.step = only_step
    if "noadam" in stage:
        tr.optimizer.step = lambda: None
    if "noreducer" in stage:
        tr.reducer.prepare = lambda: [p.__setattr__("grad", None) for p in tr.params]
        tr.reducer.finish = lambda: None
    gen = torch.Generator().manual_seed(17)
    rgbs = torch.rand(F, H, H, 3, generator=gen).to(dev)
    alphas = (torch.rand(F, H, H, 1, generator=gen) > 0.5).float().to(dev)
    fg = (torch.rand(F, 128, 3, generator=gen) * 0.4 - 0.2).to(dev) if "nopriors" not in stage else None
    bg = (torch.rand(F, 128, 3, generator=gen) * 2 - 1).to(dev) if "nopriors" not in stage else None
    fidx = (torch.arange(F) * 2 + 1).to(dev)
    use_table = tr.body_model_params is not None
    if "posegrad" in stage:
        pose = {k: v.clone().requires_grad_(True) for k, v in pose.items()}
    if "indexsel" in stage:
        table.forward = lambda ids: {n: getattr(table, n).weight.index_select(0, torch.zeros_like(ids) if n == "betas" else ids)
                                     for n in table.param_names}

    def trainer_body():
        return tr._step_body(rays, rgbs, alphas, None if use_table else pose, templ, fg, bg, 0.0 if "noperturb" in stage else 1.0,
                             fidx if use_table else None)[0]


def bump():
    if "bump" in full_stage:
        from anim_nerf_amd.autograd import bump_generation
        bump_generation([p for p in m.parameters()])


if "sg" in stage:
    for it in range(6):
        loss, _ = tr.step_graphed(rays, rgbs, alphas, None if use_table else pose, templ, fg, bg, perturb=0.0 if "noperturb" in stage else 1.0,
                                  frame_idx=fidx if use_table else None)
        print(it, float(loss), tr._graph is not None, flush=True)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(50):
        loss, _ = tr.step_graphed(rays, rgbs, alphas, None if use_table else pose, templ, fg, bg, perturb=0.0 if "noperturb" in stage else 1.0,
                                  frame_idx=fidx if use_table else None)
    torch.cuda.synchronize()
    print("ms per graphed step", (time.perf_counter() - t0) / 50 * 1e3, float(loss))
    if os.environ.get("HAZARD"):
        torch.cuda.synchronize()
        torch.ones(3, device=dev).sum().item()
        for it in range(5):
            loss, _ = tr.step_graphed(rays, rgbs, alphas, None if use_table else pose, templ, fg, bg, perturb=0.0 if "noperturb" in stage else 1.0,
                                      frame_idx=fidx if use_table else None)
        torch.cuda.synchronize()
        print(stage, "hazard sequence survived", float(loss), flush=True)
    raise SystemExit(0)
if os.environ.get("WITH_TRAINER"):       # a Trainer exists (hooks, sinks, flat buffers, its stream) but the body does not go through it
    tr = ana.Trainer(m, vr, ana.TrainHParams(n_samples=32, n_importance=16, lambda_normals=0.0), graph=True)
    if os.environ["WITH_TRAINER"] == "prepare":
        _body0 = body
        def body():
            tr.begin_step()
            return _body0()
side = tr._stream if ((stage.startswith("trainer") or os.environ.get("WITH_TRAINER")) and tr._stream is not None) else torch.cuda.Stream()
pre = os.environ.get("PRE", "")
if "alloc" in pre:                       # a persistent allocation made on the capture stream before anything else
    with torch.cuda.stream(side):
        keep_alive = torch.zeros(1 << 22, device=dev)
if "wait" in pre:                        # the default stream has waited for the capture stream once
    torch.cuda.current_stream().wait_stream(side)
if "hooks" in pre:                       # post-accumulate-grad hooks on every parameter (they keep the AccumulateGrad nodes alive)
    with torch.cuda.stream(side):
        for p in m.parameters():
            if p.requires_grad:
                p.register_post_accumulate_grad_hook(lambda q: None)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        ref = body()
        bump()
        if not stage.startswith("trainer"):
            for p in m.parameters():
                p.grad = None
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
print(stage, "capturing", flush=True)
with torch.cuda.graph(g, stream=side):
    out = body()
print(stage, "captured", flush=True)
g.replay()
torch.cuda.synchronize()
if os.environ.get("HAZARD"):
    # the sequence of tools/soak_train.py: replays on a side stream, device synchronise, work on the default stream, replays
    for _cycle in range(int(os.environ.get("HAZARD_CYCLES", "1"))):
        with torch.cuda.stream(side):
            for _ in range(int(os.environ["HAZARD"])):
                g.replay()
        torch.cuda.synchronize()
        torch.ones(3, device=dev).sum().item()
        with torch.cuda.stream(side):
            for _ in range(5):
                g.replay()
        torch.cuda.synchronize()
    print(stage, "hazard sequence survived", flush=True)
print(stage, "replayed OK", float(out.float().abs().sum()), float(ref.float().abs().sum()), flush=True)
