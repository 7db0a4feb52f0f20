#!/usr/bin/env python3
"""tests/golden/mlp_twin.npz: outputs of the reference's models/mlp.py NeRF (the pre-embedded twin of models/nerf.py's,
models/mlp.py:226-297) on seeded inputs.  Build container only:  python tests/golden/make_mlp_twin_fixture.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf                                        # noqa: E402

mf.import_reference()
import models.mlp as r_mlp                                        # noqa: E402
import models.embedding as r_emb                                  # noqa: E402

out = {}
g = torch.Generator().manual_seed(77)
xyz = torch.cat([torch.rand(1200, 3, generator=g) * 2 - 1, torch.randn(336, 3, generator=g) * 2], 0)
vd = torch.nn.functional.normalize(torch.randn(1536, 3, generator=g), dim=-1)
emb_xyz, emb_dir = r_emb.Embedding(3, 10)(xyz), r_emb.Embedding(3, 4)(vd)
# an input that is NOT the embedding of its first three channels: the kernel must take the 63 channels as they come
free = torch.randn(256, 63, generator=g) * 0.7
for tag, dirs in (("view", 27), ("plain", 0)):
    torch.manual_seed(91)
    net = r_mlp.NeRF(in_channels_dir=dirs).eval()
    with torch.no_grad():
        if dirs:
            rgb, sig = net(emb_xyz, emb_dir)
            rgb2, sig2 = net(free, emb_dir[:256])
        else:
            rgb, sig = net(emb_xyz, emb_xyz[:, :0])
            rgb2, sig2 = net(free, free[:, :0])
        so = net(emb_xyz, only_sigma=True)
    out.update({f"{tag}_rgb": rgb.numpy(), f"{tag}_sigma": sig.numpy(), f"{tag}_rgb_free": rgb2.numpy(),
                f"{tag}_sigma_free": sig2.numpy(), f"{tag}_only_sigma": so.numpy(),
                f"{tag}_weights_abs_sum": np.float64(sum(p.detach().double().abs().sum() for p in net.parameters()))})
np.savez_compressed(os.path.join(mf.OUT, "mlp_twin.npz"), seed=91, xyz=xyz.numpy(), viewdir=vd.numpy(), free=free.numpy(), **out)
print("mlp_twin.npz written", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})
