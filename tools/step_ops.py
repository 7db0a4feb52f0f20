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
# A dispatch mode sees every aten op with the Python stack that issued it; autograd's device threads are switched off so that
# the backward pass (custom Function.backward bodies and autograd's own accumulations) runs under the mode too.
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

PKG = os.sep + "anim-nerf_amd" + os.sep
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.sep
NO_LAUNCH = {"aten.empty.memory_format", "aten.empty_strided.default", "aten.view.default", "aten._unsafe_view.default", "aten.as_strided.default",
             "aten.detach.default", "aten.alias.default", "aten.slice.Tensor", "aten.select.int", "aten.expand.default", "aten.t.default",
             "aten.transpose.int", "aten.permute.default", "aten.unsqueeze.default", "aten.squeeze.dim", "aten.reshape.default",
             "aten.empty_like.default", "aten.new_empty.default", "aten.lift_fresh.default", "aten.unbind.int", "aten.split.Tensor",
             "aten._local_scalar_dense.default", "aten.view_as.default", "aten.unflatten.int", "aten.flatten.using_ints",
             "aten.squeeze.default", "aten.narrow.default", "aten.is_same_size.default", "aten.split_with_sizes.default",
             "aten.new_empty_strided.default", "aten.resize_.default", "aten.set_.source_Storage_storage_offset", "aten.unsafe_split.Tensor",
             "aten._reshape_alias.default", "aten.movedim.int", "aten.chunk.default", "aten.result_type.Tensor", "aten.sym_size.int",
             "aten.record_stream.default", "aten.is_pinned.default", "aten.unfold.default", "aten.diagonal.default", "aten.real.default"}


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.shape = {}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        if name in NO_LAUNCH:
            return out
        t = next((a for a in args if torch.is_tensor(a)), None)
        if t is not None and not t.is_cuda and not (torch.is_tensor(out) and out.is_cuda):
            return out                                              # host-side arithmetic
        frames = [f for f in traceback.extract_stack() if PKG in f.filename]
        where = f"{frames[-1].filename.replace(ROOT, '')}:{frames[-1].lineno} {frames[-1].name}" if frames else "(autograd engine / torch internals)"
        outer = f" <- {frames[-2].filename.replace(ROOT, '').split(os.sep)[-1]}:{frames[-2].lineno}" if len(frames) > 1 else ""
        key = (where + outer, name)
        self.rows[key] += 1
        self.shape.setdefault(key, tuple(t.shape) if t is not None else ())
        return out


torch.autograd.set_multithreading_enabled(False)
log = Log()
with log:
    step()
torch.cuda.synchronize()
total = sum(log.rows.values())
print(f"cfg4 eager step, {F} frames: {total} framework ops that launch (or copy), by call site")
for (where, name), n in sorted(log.rows.items(), key=lambda kv: kv[0]):
    print(f"{n:4d}  {name:38s} {str(log.shape[(where, name)]):22s} {where}")
