#!/usr/bin/env python3
"""The fused coarse pass (anr_composite_sample: compositing of 64 coarse samples + importance sampling + merge) on 2^20 rays,
as configs[1] runs it (stratified depths from the step table, every sample valid) and as configs[2] runs it (depths from
memory, validity bytes, ~12 % of the samples valid), and the masked fine compositor of configs[2].
    python tools/bench_composite_sample.py [reps=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anim_nerf_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
R, Kc, Kf = 1 << 20, 64, 64
g = torch.Generator(device=dev).manual_seed(0)
rgbs = torch.rand(R, Kc, 4, device=dev, generator=g)
rgbs[..., 3] = (torch.rand(R, Kc, device=dev, generator=g) - 0.3) * 40
rays = torch.zeros(R, 8, device=dev); rays[:, 6] = 0.5; rays[:, 7] = 2.5
steps = torch.linspace(0, 1, Kc, device=dev)
z = (rays[:, 6:7] * (1 - steps) + rays[:, 7:8] * steps).contiguous()
u = torch.linspace(0, 1, Kf, device=dev)
# validity: a run of ~11 samples along the rays that hit the body — an ellipse over ~30 % of the 1024 x 1024 image, as on a
# configs[2] frame (neighbouring rays share their fate: most wavefronts are all-hit or all-miss)
start = torch.randint(0, Kc - 11, (R, 1), device=dev, generator=g)
k = torch.arange(Kc, device=dev)[None]
yy, xx = torch.meshgrid(torch.arange(1024, device=dev), torch.arange(1024, device=dev), indexing="ij")
hit = ((((xx - 512) / 230.0) ** 2 + ((yy - 512) / 430.0) ** 2) < 1.0).reshape(R, 1)
valid = ((k >= start) & (k < start + 11) & hit).to(torch.uint8).contiguous()

def timed(fn, nbytes):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, nbytes / ms / 1e9


out_b = R * (12 + 8 + 4 * (Kc + Kf) + (Kc + Kf))
ms, tb = timed(lambda: ops.composite_sample(rgbs, rays, u, True, steps=steps, want_perm=True), R * (16 * Kc + 8) + out_b)
print(f"composite_sample, configs[1] (steps, all valid):      {ms:.3f} ms  {tb:.2f} TB/s algorithmic")
nv = int(valid.sum())
ms, tb = timed(lambda: ops.composite_sample(rgbs, rays, u, True, z=z, valid=valid, want_perm=True), R * (4 * Kc + Kc + 8) + 16 * nv + out_b)
print(f"composite_sample, configs[2] (z, {100.0 * nv / (R * Kc):.1f} % valid):        {ms:.3f} ms  {tb:.2f} TB/s algorithmic")
K = Kc + Kf
rg = torch.rand(R, K, 4, device=dev, generator=g)
zf = torch.sort(torch.rand(R, K, device=dev, generator=g) * 2 + 0.5, -1)[0].contiguous()
vf = valid.repeat(1, 2).contiguous()
ms, tb = timed(lambda: ops.composite(rg, zf, rays, True, valid=vf, want_weights=False), R * (4 * K + K + 8 + 20) + 16 * 2 * nv)
print(f"composite (fine, 128 samples, masked):                {ms:.3f} ms  {tb:.2f} TB/s algorithmic")
ms, tb = timed(lambda: ops.composite(rg, zf, rays, True, want_weights=False), R * (4 * K + 16 * K + 8 + 20))
print(f"composite (fine, 128 samples, all valid):             {ms:.3f} ms  {tb:.2f} TB/s algorithmic")
