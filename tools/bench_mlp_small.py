#!/usr/bin/env python3
"""Training MLP passes at small listed-row counts (the per-rank batch of the reference's 8-GPU run: 17-25 k rows per pass):
forward+save, activation gradients, sigma-only variants — with the row count on the device inside a large buffer, as the
explicit step calls them.  Rows <= 32,768 run in HALF TILES (csrc/mlp_core.h: Mlp::HALFABLE); the script also checks that a
row's results do not depend on the tile shape (count = n against count = 40,000 on the same buffer: bit for bit).
A/B: ANIMNERF_HIP_LIB=<library built with -DANR_HALF_TILES=0> python tools/bench_mlp_small.py"""
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
mode = ops.MLP_MODES[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
pack, bpack = ops.mlp_pack(P, mode), ops.mlp_pack(P, mode, backward=True)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
N = 131072
pts = torch.cat([torch.rand(N, 3, device=dev) * 2 - 1, torch.ones(N, 1, device=dev)], -1)
g4 = torch.randn(N, 4, device=dev)
def cnt(n): return torch.tensor([n], dtype=torch.int32, device=dev)
ref_c = cnt(40000)
out_r, act_r = ops.mlp_forward_save(pack, mode, pts, count=ref_c)
dact_r = ops.mlp_backward(bpack, mode, g4, act_r, count=ref_c)
for n in (1000, 4096, 17000, 25000, 32768, 32832, 40000, 65536, 131072):
    c = cnt(n)
    out, act = ops.mlp_forward_save(pack, mode, pts, count=c)
    dact = ops.mlp_backward(bpack, mode, g4, act, count=c)
    same = ""
    if n <= 40000:
        ok = (torch.equal(out[:n], out_r[:n]) and torch.equal(ops.act_columns(act)[:n], ops.act_columns(act_r)[:n])
              and torch.equal(ops.act_columns(dact)[:n], ops.act_columns(dact_r)[:n]))
        same = "  == the 40,000-row call on its rows" if ok else "  DIFFERS from the 40,000-row call"
    r = dict(save=t(lambda: ops.mlp_forward_save(pack, mode, pts, count=c)),
             bits=t(lambda: ops.mlp_forward_save(pack, mode, pts, count=c, bits_only=True)),
             bwd=t(lambda: ops.mlp_backward(bpack, mode, g4, act, count=c)),
             bwd_enc=t(lambda: ops.mlp_backward(bpack, mode, g4, act, count=c, enc_only=True)))
    print(f"rows={n:7d} " + "  ".join(f"{k} {v:7.1f} us" for k, v in r.items()) + same)
