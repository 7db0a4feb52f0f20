"""`models/mlp.py`'s NeRF (the pre-embedded twin of `models/nerf.py`'s, models/mlp.py:226-297): the caller embeds
positions and directions itself and hands the 63- / 27-channel vectors over.  No caller in the reference uses it, but
BASELINE.json's north_star names the file; constructor, state-dict keys and `forward(input_xyz, input_dir,
only_sigma)` are the reference's, the trunk runs in the same fused HIP kernel with its encoder skipped
(`anr_mlp_forward_embedded`).

With `in_channels_dir == 0` the whole network is one launch.  With view-direction channels (the class default, 27) the
colour head `[feature, input_dir] -> 128 -> 3` has 27 more inputs than the fused kernel's: trunk, sigma and the
256-wide feature come from the kernel, the head runs as two library GEMMs — inference only, like `nerf.NeRF(use_view=True)`.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .nerf import _default_mode


class NeRF(nn.Module):
    def __init__(self, D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=[4], mlp_mode: Optional[str] = None):
        super().__init__()
        self.D, self.W = D, W
        self.in_channels_xyz, self.in_channels_dir = in_channels_xyz, in_channels_dir
        self.skips = skips
        self.mlp_mode = mlp_mode or _default_mode()
        for i in range(D):
            fan_in = in_channels_xyz if i == 0 else W + in_channels_xyz if i in skips else W
            setattr(self, f"xyz_encoding_{i+1}", nn.Sequential(nn.Linear(fan_in, W), nn.ReLU(True)))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.dir_encoding = nn.Sequential(nn.Linear(W + in_channels_dir, W // 2), nn.ReLU(True))
        self.sigma = nn.Linear(W, 1)
        self.rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())
        self._pack_cache = {}

    def _supported(self):
        return self.D == 8 and self.W == 256 and self.in_channels_xyz == 63 and list(self.skips) == [4]

    def weight_pack(self, mode: Optional[str] = None):
        if not self._supported():
            raise NotImplementedError("HIP MLP covers D=8, W=256, in_channels_xyz=63, skips=[4]")
        mode_id = ops.MLP_MODES[mode or self.mlp_mode]
        params = dict(self.named_parameters())
        key = (mode_id, tuple((p.data_ptr(), p._version) for p in params.values()))
        hit = self._pack_cache.get(mode_id)
        if hit is None or hit[0] != key:
            if self.in_channels_dir:          # the kernel's own colour head is not used then: give it the feature columns
                params["dir_encoding.0.weight"] = params["dir_encoding.0.weight"][:, :self.W].contiguous()
            hit = (key, ops.mlp_pack(params, mode_id))
            self._pack_cache[mode_id] = hit
        return hit[1], mode_id

    def forward(self, input_xyz, input_dir=None, only_sigma=False):
        """models/mlp.py:268-297: input_xyz[..., 63], input_dir[..., in_channels_dir] -> sigma[..., 1] if only_sigma else
        (rgb[..., 3], sigma[..., 1])."""
        if torch.is_grad_enabled() and (input_xyz.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("models/mlp.py's NeRF is served for inference (call under torch.no_grad()); training "
                                      "goes through models/nerf.py's NeRF, which every caller of the reference uses")
        lead = input_xyz.shape[:-1]
        emb = input_xyz.reshape(-1, self.in_channels_xyz).float().contiguous()
        pack, mode_id = self.weight_pack()
        with torch.no_grad():
            if only_sigma:
                return ops.mlp_forward_embedded(pack, mode_id, emb, sigma_only=True).view(*lead, 1)
            if self.in_channels_dir == 0:
                out = ops.mlp_forward_embedded(pack, mode_id, emb)
                return out[:, :3].reshape(*lead, 3), out[:, 3:4].reshape(*lead, 1)
            # view-dependent head (the class default): sigma and the 256-wide xyz_encoding_final feature from the kernel's
            # saved activations, then [feature, input_dir] -> 128 -> 3 as two library GEMMs
            sig, feat = [], []
            for i in range(0, emb.shape[0], 1 << 20):            # 4.9 KB of saved activations per point
                out, act = ops.mlp_forward_embedded(pack, mode_id, emb[i:i + (1 << 20)], want_act=True)
                sig.append(out[:, 3].clone())
                feat.append(ops.act_columns(act, 2048, 2304).float())
            x = torch.cat([torch.cat(feat), input_dir.reshape(-1, self.in_channels_dir).float()], -1)
            rgb = self.rgb(self.dir_encoding(x))
            return rgb.view(*lead, 3), torch.cat(sig).view(*lead, 1)
