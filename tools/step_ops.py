#!/usr/bin/env python3
"""Which FRAMEWORK ops (aten::*, each one or more kernel launches) does one eager cfg4 training step issue, and from which line
of the package?  torch.profiler with Python stacks; ops grouped by their innermost frame inside anim-nerf_amd/.  The library's
own kernels go through ctypes and do not show here: everything listed is a launch to fold into a kernel of ours.
    python tools/step_ops.py [frames_per_gpu]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
torch.manual_seed(0)
model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True, use_fine=True, mlp_mode="bf16").to(dev)
hp = ana.TrainHParams(n_samples=64, n_importance=32, chunk=2048)
table = ana.BodyModelParams(114).to(dev)
seeded = syn.animated_pose_params(seed=200, bs=114)
for name in table.param_names:
    table.init_parameters(name, torch.from_numpy(seeded[name]).to(dev), requires_grad=True)
trainer = ana.Trainer(model, ana.VolumeRenderer(n_coarse=64, n_fine=32), hp, body_model_params=table)
frame_idx = torch.arange(F, device=dev) * (114 // F)
c2w, focal, cen = syn.pinhole_camera(32, 32)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 32, 32, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(F, 1, 1, 1)
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
g = torch.Generator().manual_seed(0)
rgbs = torch.rand(F, 32, 32, 3, generator=g).to(dev)
alphas = (torch.rand(F, 32, 32, 1, generator=g) > 0.5).float().to(dev)
fg = (torch.rand(F, 128, 3, generator=g) * 0.4 - 0.2).to(dev)
bg = (torch.rand(F, 128, 3, generator=g) * 2 - 1).to(dev)
step = lambda: trainer.step(rays, rgbs, alphas, None, templ, fg, bg, perturb=1.0, frame_idx=frame_idx)
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()

# leaf aten ops only (an aten::zeros contains aten::empty + aten::zero_ + aten::fill_: count the one that launches)
LAUNCHING = {"aten::fill_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::sub", "aten::div", "aten::div_",
             "aten::cat", "aten::index_select", "aten::embedding", "aten::index", "aten::index_put_", "aten::sum", "aten::mean",
             "aten::exp", "aten::log10", "aten::where", "aten::gt", "aten::lt", "aten::neg", "aten::mse_loss", "aten::normal_",
             "aten::uniform_", "aten::_to_copy", "aten::clone", "aten::zero_", "aten::embedding_dense_backward", "aten::sqrt",
             "aten::pow", "aten::rsub", "aten::relu", "aten::threshold_backward", "aten::mm", "aten::bmm", "aten::addmm",
             "aten::_foreach_add_", "aten::select_backward", "aten::slice_backward", "aten::index_add_", "aten::scatter_add_",
             "aten::randn_like", "aten::rand", "aten::randn", "aten::max", "aten::min", "aten::arange", "aten::linspace", "aten::sort",
             "aten::cumsum", "aten::norm", "aten::linalg_vector_norm", "aten::eq", "aten::ne", "aten::bitwise_and", "aten::any"}
PKG = os.sep + "anim-nerf_amd" + os.sep
rows = collections.Counter()
shapes = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.name not in LAUNCHING:
        continue
    # skip ops nested inside another launching op (copy_ inside _to_copy / clone, fill_ inside zero_ ...)
    parent, nested = ev.cpu_parent, False
    while parent is not None:
        if parent.name in LAUNCHING:
            nested = True
            break
        parent = parent.cpu_parent
    if nested:
        continue
    where = next((f for f in (ev.stack or []) if PKG in f), None)
    if where is None:
        where = next((f for f in (ev.stack or []) if "tools" in f or "torch/autograd" in f or "optim" in f), "(no package frame: autograd engine / optimiser)")
    where = where.replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.sep, "")
    rows[(where, ev.name)] += 1
    shapes.setdefault((where, ev.name), str(ev.input_shapes)[:70])
total = sum(rows.values())
print(f"cfg4 eager step, {F} frames: {total} launching framework ops")
for (where, name), n in sorted(rows.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    print(f"{n:4d}  {name:28s} {where}   {shapes[(where, name)]}")
