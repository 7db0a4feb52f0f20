#!/usr/bin/env python3
"""Micro-benchmark of anr_mlp_forward alone: variants interleaved in one process (A/B within one probe).
Usage: python tools/bench_mlp.py [n_points] [rounds] [modes,comma,separated]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
modes = (sys.argv[3] if len(sys.argv) > 3 else "bf16,bf16_w4,f32").split(",")
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ana.NeRF(freqs_dir=0, use_view=False).to(dev)
pts = torch.cat([torch.rand(n, 3, device=dev) * 2 - 1, torch.ones(n, 1, device=dev)], -1)
FLOP = 1_179_904
res = {m: [] for m in modes}
ref = None
for r in range(rounds + 1):
    for m in modes:
        pack, mode = net.weight_pack(m.split("_")[0])
        mode = ops.MLP_MODES[m]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = ops.mlp_forward(pack, mode, pts)
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            res[m].append(e0.elapsed_time(e1))
        if r == 0:
            if m == "f32":
                ref = out
for m in modes:
    t = sorted(res[m])
    med, best = t[len(t) // 2], t[0]
    peak = 157.3 if m.startswith("f32") else 2500.0
    print(f"{m:12s} n={n} median {med:8.3f} ms  best {best:8.3f} ms  -> {n*FLOP/med/1e9:8.1f} TFLOP/s median ({n*FLOP/med/1e9/peak*100:5.1f}% of {peak})")
