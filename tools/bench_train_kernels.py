#!/usr/bin/env python3
"""Per-kernel times of the training MLP path at several row counts: forward, forward+save, activation gradients, weight gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import ops
from anim_nerf_amd.autograd import PARAM_KEYS
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = ana.NeRF(freqs_dir=0, use_view=False).to(dev)
named = dict(net.named_parameters())
P = {k: named[k].detach() for k in PARAM_KEYS}
mode = ops.MLP_MODES["bf16"]
pack, bpack = ops.mlp_pack(P, mode), ops.mlp_pack(P, mode, backward=True)
FLOP = 1_179_904
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n in (65536, 131072, 1 << 20, 1 << 22):
    pts = torch.cat([torch.rand(n, 3, device=dev) * 2 - 1, torch.ones(n, 1, device=dev)], -1)
    g4 = torch.randn(n, 4, device=dev)
    out, act = ops.mlp_forward_save(pack, mode, pts)
    dact = ops.mlp_backward(bpack, mode, g4, act)
    enc = ops.encode64(pts, act.dtype)
    r = dict(fwd=t(lambda: ops.mlp_forward(pack, mode, pts)), save=t(lambda: ops.mlp_forward_save(pack, mode, pts)),
             bwd=t(lambda: ops.mlp_backward(bpack, mode, g4, act)), wgrad=t(lambda: ops.mlp_wgrad(mode, act, dact, enc, g4)))
    print(f"n={n:8d} " + "  ".join(f"{k} {v:7.3f} ms ({n*FLOP/v/1e9/2500*100:5.1f}% peak)" for k, v in r.items()))
