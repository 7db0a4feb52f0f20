#!/usr/bin/env python3
"""tests/golden/mlp_view.npz: the reference's NeRF with use_view=True (its class default; no shipped config) evaluated
on seeded points and view directions — RUNS THE REFERENCE (imported from /root/reference), stores inputs/outputs only.

  python tests/golden/make_view_fixture.py
"""
import os, sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from models.nerf import NeRF            # noqa: E402

SEED = 21
torch.manual_seed(SEED)
net = NeRF(freqs_xyz=10, freqs_dir=4, use_view=True)        # layer creation order = ours: same seeded weights
g = torch.Generator().manual_seed(5)
xyz = torch.rand(1, 300, 3, generator=g) * 2 - 1
d = torch.randn(1, 300, 3, generator=g)
viewdir = d / d.norm(dim=-1, keepdim=True)
with torch.no_grad():
    rgb, sigma = net(xyz, viewdir)
    sig2, feat = net.get_sigma(xyz)
chk = float(sum(p.double().abs().sum() for p in net.parameters()))
np.savez_compressed(os.path.join(HERE, "mlp_view.npz"), seed=SEED, xyz=xyz.numpy(), viewdir=viewdir.numpy(), rgb=rgb.numpy(),
                    sigma=sigma.numpy(), feature=feat.numpy(), weights_abs_sum=chk)
print("wrote mlp_view.npz", rgb.shape, sigma.shape, feat.shape, chk)
