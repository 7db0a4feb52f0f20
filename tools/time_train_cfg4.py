#!/usr/bin/env python3
"""Wall time per phase of the bench's cfg4 training step (pose refinement on, normals term on), with a device
synchronisation between phases, plus the number of kernel launches in each phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
from anim_nerf_amd.render import system_forward
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
torch.manual_seed(0)
model = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=True, use_knn=True, use_fine=True, mlp_mode="bf16").to(dev)
hp = ana.TrainHParams(n_samples=64, n_importance=32, chunk=2048)
if len(sys.argv) > 1 and sys.argv[1] == "nonormals":
    hp.lambda_normals = 0.0
F = 16
table = ana.BodyModelParams(114).to(dev)
seeded = syn.animated_pose_params(seed=200, bs=114)
for name in table.param_names:
    table.init_parameters(name, torch.from_numpy(seeded[name]).to(dev), requires_grad=True)
tr = ana.Trainer(model, ana.VolumeRenderer(n_coarse=64, n_fine=32), hp, body_model_params=table)
frame_idx = torch.arange(F, device=dev) * (114 // F)
c2w, focal, cen = syn.pinhole_camera(32, 32)
rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 32, 32, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(F, 1, 1, 1)
templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
g = torch.Generator().manual_seed(0)
rgbs = torch.rand(F, 32, 32, 3, generator=g).to(dev); alphas = (torch.rand(F, 32, 32, 1, generator=g) > 0.5).float().to(dev)
fg = (torch.rand(F, 128, 3, generator=g) * 0.4 - 0.2).to(dev); bg = (torch.rand(F, 128, 3, generator=g) * 2 - 1).to(dev)
def sync(): torch.cuda.synchronize(); return time.perf_counter()

def one(count=False):
    marks = []
    def phase(name, fn):
        t0 = sync()
        if count:
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                out = fn(); torch.cuda.synchronize()
            evs = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
            n = sum(e.count for e in evs)
            if os.environ.get("ANR_LAUNCH_NAMES"):
                print(f"--- {name}: {n} launches")
                for e in sorted(evs, key=lambda e: -e.count)[:40]:
                    print(f"   {e.count:4d} x {e.key[:110]}  ({e.device_time_total / max(e.count, 1):.1f} us)")
        else:
            out = fn(); n = -1
        marks.append((name, 1e3 * (sync() - t0), n))
        return out
    tr.begin_step()
    pose = phase("table", lambda: table(frame_idx))
    phase("smpl", lambda: model.set_body_model(pose, templ))
    flat = phase("frame-prep", lambda: (model.convert_to_body_model_space(rays.view(F, 1024, 8)), model.clac_ober2cano_transform(), model.knn_index())[0])
    res = phase("render-fwd", lambda: ana.render_prepared(tr.renderer, model, flat, chunk=2048, perturb=1.0))
    res = {k: v.view(F, 32, 32, -1) for k, v in res.items()}
    loss = phase("loss", lambda: ana.compute_loss(model, hp, rgbs, alphas, res, fg, bg)[0])
    phase("backward", lambda: loss.backward())
    phase("adam", lambda: tr.optimizer.step())
    return marks

for it in range(3):
    one()
for count in (False, True):
    m = one(count)
    print("  ".join(f"{n} {t:6.1f}" + (f" ({k} launches)" if k >= 0 else "") for n, t, k in m), " total %.1f ms" % sum(t for _, t, _ in m))
t0 = sync()
for _ in range(5):
    tr.step(rays, rgbs, alphas, None, templ, fg, bg, perturb=1.0, frame_idx=frame_idx)
print("unsynchronised step: %.1f ms" % (1e3 * (sync() - t0) / 5))
