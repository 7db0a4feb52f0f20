"""`Embedding` and `NeRF` with the reference's constructor arguments and state-dict keys
(models/embedding.py:5-39, models/nerf.py:60-190), evaluated by the fused HIP MLP kernel.

Layers are created in the reference's order (trunk 1..8, final, dir, sigma, rgb), so
`torch.manual_seed(s)` followed by construction gives the same random-init weights as the
reference, and a reference checkpoint loads with `load_state_dict` unchanged.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops


class Embedding(nn.Module):
    """x -> (x, sin(2^k x), cos(2^k x))_k.  Stand-alone use is not on the hot path (the MLP
    kernel fuses the encoding); this is the plain tensor-op form of models/embedding.py:22-39."""

    def __init__(self, in_channels: int, N_freqs: int, logscale: bool = True):
        super().__init__()
        self.N_freqs = N_freqs
        self.in_channels = in_channels
        self.out_channels = in_channels * (2 * N_freqs + 1)
        self.freq_bands = (2 ** torch.linspace(0, N_freqs - 1, N_freqs) if logscale
                           else torch.linspace(1, 2 ** (N_freqs - 1), N_freqs))

    def forward(self, x):
        cols = [x]
        for f in self.freq_bands.tolist():
            cols += [torch.sin(f * x), torch.cos(f * x)]
        return torch.cat(cols, -1)


def _default_mode() -> str:
    return os.environ.get("ANIMNERF_MLP_MODE", "f32")


class NeRF(nn.Module):
    def __init__(self, D=8, W=256, freqs_xyz=10, freqs_dir=4, use_view=True, use_normal=False,
                 deformation_dim=0, apperance_dim=0, skips=[4], actvn_type="relu", mlp_mode: Optional[str] = None):
        super().__init__()
        self.D, self.W = D, W
        self.freqs_xyz, self.freqs_dir = freqs_xyz, freqs_dir
        self.deformation_dim, self.apperance_dim = deformation_dim, apperance_dim
        self.skips = skips
        self.use_view, self.use_normal = use_view, use_normal
        self.mlp_mode = mlp_mode or _default_mode()

        self.encoding_xyz = Embedding(3, freqs_xyz)
        if use_view:
            self.encoding_dir = Embedding(3, freqs_dir)
        self.in_channels_xyz = 3 + 6 * freqs_xyz + deformation_dim
        self.in_channels_dir = apperance_dim + (3 + 6 * freqs_dir if use_view else 0) + (3 if use_normal else 0)
        if actvn_type != "relu":
            raise NotImplementedError("the HIP MLP implements actvn_type='relu' (every shipped config)")

        for i in range(D):
            fan_in = self.in_channels_xyz if i == 0 else W + self.in_channels_xyz if i in skips else W
            setattr(self, f"xyz_encoding_{i+1}", nn.Sequential(nn.Linear(fan_in, W), nn.ReLU(inplace=True)))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.dir_encoding = nn.Sequential(nn.Linear(W + self.in_channels_dir, W // 2), nn.ReLU(True))
        self.sigma = nn.Linear(W, 1)
        self.rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())
        self._pack_cache = {}
        self.grad_sink = None            # autograd.GradSink, attached by the trainer: weight gradients accumulate in one flat buffer

    # -- weight pack (fragment-ordered copy for the kernel), rebuilt when any parameter changes
    def _trunk_supported(self):
        return (self.D == 8 and self.W == 256 and self.freqs_xyz == 10 and list(self.skips) == [4]
                and not self.use_normal and self.deformation_dim == 0 and self.apperance_dim == 0)

    def _hip_supported(self):
        """The whole network in the fused kernel (every shipped config); use_view=True runs its colour head outside."""
        return self._trunk_supported() and not self.use_view

    def _view_fused(self):
        """the view-dependent colour head inside the fused kernel: [feature, Embedding(viewdir)] -> 128 -> 3 and nothing else
        in the head's input (no normal input, no appearance code)"""
        return self._trunk_supported() and self.use_view and self.in_channels_dir == 3 + 6 * self.freqs_dir and self.freqs_dir <= 10

    def weight_pack(self, mode: Optional[str] = None, view: bool = False):
        if not self._trunk_supported():
            raise NotImplementedError(
                "HIP MLP covers D=8, W=256, freqs_xyz=10, skips=[4], no normal input, no latent codes (configs/**/*.yaml)")
        mode_id = ops.MLP_MODES[mode or self.mlp_mode]
        from .autograd import weights_generation
        params = {k: v for k, v in self.named_parameters()}
        # (the generation: an optimiser step that does not bump the version counters — torch's fused Adam — still repacks)
        key = (mode_id, weights_generation(params["xyz_encoding_1.0.weight"]), tuple((p.data_ptr(), p._version) for p in params.values()))
        slot = (mode_id, view)
        hit = self._pack_cache.get(slot)
        if hit is None or hit[0] != key:
            if view:                # the whole network, view-dependent head included (inference)
                hit = (key, ops.mlp_pack(params, mode_id, view_channels=self.in_channels_dir))
            else:
                if self.use_view:   # the kernel's own colour head is not used then: give it the 256 feature columns
                    params = dict(params)
                    params["dir_encoding.0.weight"] = params["dir_encoding.0.weight"][:, :self.W].contiguous()
                hit = (key, ops.mlp_pack(params, mode_id))
            self._pack_cache[slot] = hit
        return hit[1], mode_id

    def _training(self, pts=None):
        return torch.is_grad_enabled() and ((pts is not None and pts.requires_grad) or any(p.requires_grad for p in self.parameters()))

    def sigma_and_feature(self, pts: torch.Tensor, mode: Optional[str] = None, chunk: int = 1 << 20):
        """(sigma[n], xyz_encoding_final[n,256]) of NeRF.get_sigma (models/nerf.py:155-175) from the training-forward
        kernel, which stores that feature among the saved activations.  Under autograd: `autograd.FeatureFunction`
        (differentiable w.r.t. the trunk / sigma / xyz_encoding_final tensors and the points; the feature's gradient
        re-enters the fused backward kernel).  `chunk` bounds the 4.9 KB per point of saved activations in inference."""
        if self._training(pts):
            from .autograd import PARAM_KEYS, FeatureFunction
            named = dict(self.named_parameters())
            return FeatureFunction.apply(pts, ops.MLP_MODES[mode or self.mlp_mode] & 0xff, *[named[k] for k in PARAM_KEYS])
        with torch.no_grad():
            pack, mode_id = self.weight_pack(mode)
            sig, feat = [], []
            for i in range(0, pts.shape[0], chunk):
                out, act = ops.mlp_forward_save(pack, mode_id, pts[i:i + chunk].contiguous())
                sig.append(out[:, 3].clone())
                feat.append(ops.act_columns(act, 2048, 2304).float())
            return torch.cat(sig), torch.cat(feat)

    def eval_points_view(self, pts: torch.Tensor, viewdir: torch.Tensor, mode: Optional[str] = None) -> torch.Tensor:
        """use_view=True (the class default of the reference, no shipped config).  Inference: the whole network — trunk, sigma,
        feature and the view-dependent colour head (models/nerf.py:141-153: [feature, encoding_dir(viewdir)] -> 128 -> 3) — in
        the fused kernel (anr_mlp_forward_view).  Under autograd: trunk, sigma and the feature in the fused kernels, the head as
        framework ops whose input gradient goes back into the fused backward (sigma_and_feature).  -> [n,4] = (r,g,b,sigma)."""
        if not self._training(pts) and self._view_fused() and pts.is_cuda:
            # inference: the whole network in the fused kernel (anr_mlp_forward_view), the direction's Fourier panel in registers
            pack, mode_id = self.weight_pack(mode, view=True)
            return ops.mlp_forward_view(pack, mode_id, pts, viewdir.reshape(-1, 3).float().contiguous())
        sig, feat = self.sigma_and_feature(pts, mode)
        with torch.set_grad_enabled(self._training(pts)):
            x = torch.cat([feat, self.encoding_dir(viewdir.reshape(-1, 3).float())], -1)
            rgb = self.rgb(self.dir_encoding(x))
            return torch.cat([rgb, sig[:, None]], -1)

    def eval_points(self, pts: torch.Tensor, mode: Optional[str] = None, sigma_only: bool = False,
                    only_valid: bool = False, valid_list=None) -> torch.Tensor:
        """pts[n,4] = (x,y,z,valid) -> [n,4] = (r,g,b,sigma), or sigma[n] (trunk + sigma row only).  The fused kernel entry.
        only_valid: run the network on the samples with valid >= 1 only; the rest get (0,0,0,-1e5)."""
        if self.use_view:
            raise NotImplementedError("use_view=True: the colour needs the view direction, call eval_points_view(pts, viewdir)")
        if torch.is_grad_enabled() and (pts.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .autograd import PARAM_KEYS, MLPFunction              # training: keep activations, differentiable
            if not self._hip_supported():
                raise NotImplementedError("HIP MLP covers the shipped configuration only")
            named = dict(self.named_parameters())
            riders = self._riders
            if riders is not None and not sigma_only and pts.shape[0] >= 8 * riders.shape[0]:
                # canonical points that ride along with this (much larger) batch: see attach_riders
                self._riders = None
                n = pts.shape[0]
                both = torch.cat([pts, torch.cat([riders, torch.ones_like(riders[:, :1])], 1)], 0)
                out = MLPFunction.apply(both, False, ops.MLP_MODES[mode or self.mlp_mode] & 0xff, only_valid, self.grad_sink,
                                        *[named[k] for k in PARAM_KEYS])
                self._rider_sigma = out[n:, 3].contiguous()
                return out[:n]
            return MLPFunction.apply(pts, sigma_only, ops.MLP_MODES[mode or self.mlp_mode] & 0xff, only_valid, self.grad_sink,
                                     *[named[k] for k in PARAM_KEYS])
        pack, mode_id = self.weight_pack(mode)
        return ops.mlp_forward(pack, mode_id, pts, sigma_only=sigma_only, only_valid=only_valid, valid_list=valid_list)

    # Riders: a small set of canonical points whose sigma a later loss term wants (the foreground / background priors of
    # train.py:262-286: 4,096 points per step).  A field query of their own costs a forward, a backward, a weight-gradient and
    # three glue launches per network, each at the ~50 us floor of streaming 1.2 MB of weights through sixteen CUs; appended to
    # the next training-mode evaluation of this network (the step's ray samples, ~10^5 rows) they cost 3 % more rows there.
    _riders = None
    _rider_sigma = None

    def attach_riders(self, xyz: Optional[torch.Tensor]):
        """xyz[...,3] (no gradient w.r.t. the points) -> evaluated with the next training-mode `eval_points` call that is at
        least 8 times their size; `take_rider_sigma()` then returns their raw sigma, once.  None detaches."""
        self._riders = None if xyz is None else xyz.detach().reshape(-1, 3)
        self._rider_sigma = None

    def take_rider_sigma(self):
        out, self._rider_sigma, self._riders = self._rider_sigma, None, None
        return out

    def eval_rays(self, rays: torch.Tensor, z: torch.Tensor, mode: Optional[str] = None) -> torch.Tensor:
        """[n,4] = (r,g,b,sigma) at the samples o + z d of rays[bs,R,>=8], z[bs,R,K] (inference, no warp): the kernel
        generates the points itself."""
        pack, mode_id = self.weight_pack(mode)
        return ops.mlp_forward_rays(pack, mode_id, rays, z)

    def _pack_xyz(self, xyz):
        flat = xyz.reshape(-1, 3)
        return torch.cat([flat, torch.ones_like(flat[:, :1])], -1)

    def forward(self, xyz, viewdir=None, deformation_code=None, apperance_code=None):
        """models/nerf.py:129-153 -> (rgb[...,3], sigma[...,1])."""
        if self.use_view:
            out = self.eval_points_view(self._pack_xyz(xyz), viewdir).view(*xyz.shape[:-1], 4)
        else:
            out = self.eval_points(self._pack_xyz(xyz)).view(*xyz.shape[:-1], 4)
        return out[..., :3], out[..., 3:4]

    def get_sigma(self, xyz, deformation_code=None, only_sigma=False):
        """models/nerf.py:155-175.  The 256-wide feature is internal to the fused kernel."""
        if not only_sigma:              # (sigma, xyz_encoding_final): inference only
            sig, feat = self.sigma_and_feature(self._pack_xyz(xyz))
            return sig.view(*xyz.shape[:-1], 1), feat.view(*xyz.shape[:-1], self.W)
        if self.use_view:
            return self.sigma_and_feature(self._pack_xyz(xyz))[0].view(*xyz.shape[:-1], 1)
        return self.eval_points(self._pack_xyz(xyz), sigma_only=True).view(*xyz.shape[:-1], 1)

    def _sigma_dense(self, xyz):
        """sigma(xyz) as a chain of library GEMMs under torch autograd (cross-check of get_normal in the tests)."""
        e = self.encoding_xyz(xyz)
        h = e
        for i in range(self.D):
            if i in self.skips:
                h = torch.cat([e, h], -1)
            lin = getattr(self, f"xyz_encoding_{i+1}")[0]
            h = torch.relu(torch.nn.functional.linear(h, lin.weight, lin.bias))
        return self.sigma(h)

    def get_normal(self, xyz, deformation_code=None, delta=0.02):
        """models/nerf.py:177-190: d alpha / d xyz, differentiable once more w.r.t. the weights (the normals regulariser,
        train.py:288-309).  Forward-mode tangents through the fused kernels with a hand-written backward
        (`autograd.NormalFunction`); the inputs are constants of the loss, as in the reference's call sites."""
        from .autograd import PARAM_KEYS, NormalFunction
        if not self._hip_supported():
            raise NotImplementedError("get_normal is built for the shipped configuration (use_view=False, D=8, W=256)")
        named = dict(self.named_parameters())
        flat = xyz.detach().reshape(-1, 3).float().contiguous()
        with torch.set_grad_enabled(True):
            n = NormalFunction.apply(flat, float(delta), ops.MLP_MODES[self.mlp_mode] & 0xff, self.grad_sink,
                                     *[named[k] for k in PARAM_KEYS])
        return n.view(*xyz.shape[:-1], 3)

    def tangent_sigma(self, xyz):
        """quads[n_pad,4] = (sigma, d sigma / d xyz) at xyz[...,3] flattened (n_pad = n rounded up to 16, zero rows after n):
        what get_normal is computed from, for the fused loss kernels (autograd.QuadSigmaFunction)."""
        from .autograd import PARAM_KEYS, QuadSigmaFunction
        if not self._hip_supported():
            raise NotImplementedError("tangent_sigma is built for the shipped configuration (use_view=False, D=8, W=256)")
        named = dict(self.named_parameters())
        flat = xyz.detach().reshape(-1, 3)
        with torch.set_grad_enabled(True):
            return QuadSigmaFunction.apply(flat, ops.MLP_MODES[self.mlp_mode] & 0xff, self.grad_sink,
                                           *[named[k] for k in PARAM_KEYS])

    def _normal_autograd(self, xyz, delta=0.02):
        """The same quantity by autograd of autograd over `_sigma_dense` (reference formulation; used by the tests)."""
        with torch.set_grad_enabled(True):
            xyz = xyz.detach().requires_grad_(True)
            alpha = 1 - torch.exp(-delta * torch.relu(self._sigma_dense(xyz)))
            return torch.autograd.grad(alpha, xyz, torch.ones_like(alpha), create_graph=True, retain_graph=True,
                                       only_inputs=True)[0]
