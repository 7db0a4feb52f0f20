"""Autograd bridges for the training path.

Forward runs on the same HIP kernels as inference (`anr_mlp_forward_save` additionally keeps each layer's
post-activation output, `anr_composite`); backward:
  * compositing: `anr_composite_backward` (HIP, one wavefront per ray);
  * MLP: `anr_mlp_backward` (the whole activation-gradient chain in one kernel) and `anr_mlp_wgrad` (all weight and bias
    gradients: split-K MFMA GEMMs over the points between the saved activations and the activation gradients);
  * normals regulariser: the same three kernels in tangent mode (`NormalFunction`).
Pose refinement (`optim_body_params`): dL/dx_c leaves the MLP backward through the encoding, `WarpFunction` routes it
into the per-vertex observation->canonical transforms (scatter-add over the 4 neighbours, whose weights carry no
gradient: KNN distances are `no_grad` in the reference, models/anim_nerf.py:158) and into the sample position
x = o' + z d'; the per-frame chain above that (closed-form inverses, SMPL/LBS) is ordinary torch autograd.
"""
from __future__ import annotations

import torch

from . import ops

PARAM_KEYS = ([k for i in range(1, 9) for k in (f"xyz_encoding_{i}.0.weight", f"xyz_encoding_{i}.0.bias")]
              + ["sigma.weight", "sigma.bias", "xyz_encoding_final.weight", "xyz_encoding_final.bias",
                 "dir_encoding.0.weight", "dir_encoding.0.bias", "rgb.0.weight", "rgb.0.bias"])


_PACKS = {}
_GEN = {}                      # id(first parameter) -> [generation, a backward pass ran since the last forward]


def weights_generation(first_param, backward: bool = False) -> int:
    """Counts optimiser steps without trusting the parameters' version counters (torch's fused Adam updates in place
    WITHOUT bumping them): the first forward pass after a backward pass starts a new generation, and every cached weight
    pack — training, backward and inference (NeRF.weight_pack) — carries the generation it was built in."""
    st = _GEN.setdefault(id(first_param), [0, False])
    if backward:
        st[1] = True
    elif st[1]:
        st[0] += 1
        st[1] = False
    if len(_GEN) > 256:
        _GEN.clear()
        _GEN[id(first_param)] = st
    return st[0]


def bump_generation(params):
    """Call after an optimiser step that may not have bumped the parameters' version counters (torch's fused Adam does
    not): raises the counters themselves, which every weight-pack cache — `_cached_pack` here, `NeRF.weight_pack` on the
    inference side — is keyed by.  `Trainer.step` does it; the call-order heuristic of `weights_generation` stays as the
    fallback for loops that do not."""
    params = [p for p in params if isinstance(p, torch.Tensor)]
    if params:
        torch.autograd.graph.increment_version(params)


def _cached_pack(params, mode_id, backward, frozen=False):
    """Fragment-ordered weight pack for the forward / activation-gradient kernels, rebuilt only when a parameter changed:
    one pack per network and optimiser step, not one per ray chunk.  Keyed by the parameter OBJECTS (weak references: a
    new tensor that happens to reuse a freed one's address and version counter must not hit), their version counters and
    the generation above.  frozen: no optimiser touches these tensors (requires_grad False, the `_refine` stage): the
    version counters alone decide — the packs of a frozen network are made once per run, not once per step."""
    import weakref
    key = (mode_id, backward, tuple(id(p) for p in params))
    ver = (tuple(p._version for p in params), "frozen" if frozen else weights_generation(params[0], backward))
    hit = _PACKS.get(key)
    if hit is None or hit[0] != ver or any(r() is not p for r, p in zip(hit[2], params)):
        if len(_PACKS) > 64:
            _PACKS.clear()
        named = dict(zip(PARAM_KEYS, params))
        if named["dir_encoding.0.weight"].shape[1] > 256:       # use_view: the kernels see the 256 feature columns only
            named["dir_encoding.0.weight"] = named["dir_encoding.0.weight"][:, :256].contiguous()
        hit = (ver, ops.mlp_pack(named, mode_id, backward=backward), [weakref.ref(p) for p in params])
        _PACKS[key] = hit
    return hit[1]


def _cached_pack_pair(params_a, params_b, mode_id, backward, frozen=False):
    """_cached_pack of two networks; when both need a new pack, one launch makes both (ops.mlp_pack_pair)."""
    import weakref
    todo = []
    for params in (params_a, params_b):
        key = (mode_id, backward, tuple(id(p) for p in params))
        ver = (tuple(p._version for p in params), "frozen" if frozen else weights_generation(params[0], backward))
        hit = _PACKS.get(key)
        todo.append((key, ver, hit is None or hit[0] != ver or any(r() is not p for r, p in zip(hit[2], params))))
    if not (todo[0][2] and todo[1][2]) or any(dict(zip(PARAM_KEYS, ps))["dir_encoding.0.weight"].shape[1] > 256 for ps in (params_a, params_b)):
        return _cached_pack(params_a, mode_id, backward, frozen), _cached_pack(params_b, mode_id, backward, frozen)
    if len(_PACKS) > 64:
        _PACKS.clear()
    pa, pb = ops.mlp_pack_pair(dict(zip(PARAM_KEYS, params_a)), dict(zip(PARAM_KEYS, params_b)), mode_id, backward=backward)
    for (key, ver, _), params, pack in zip(todo, (params_a, params_b), (pa, pb)):
        _PACKS[key] = (ver, pack, [weakref.ref(p) for p in params])
    return pa, pb


PARAM_SHAPES = ([s for i in range(8) for s in ((256, 63 if i == 0 else 319 if i == 4 else 256), (256,))]
                + [(1, 256), (1,), (256, 256), (256,), (128, 256), (128,), (3, 128), (3,)])


class FeatureFunction(torch.autograd.Function):
    """(sigma[n], feature[n,256]) = NeRF.get_sigma(xyz) (models/nerf.py:155-175) at pts[n,4] = (x, y, z, valid): trunk, sigma
    and xyz_encoding_final in the fused kernels, differentiable w.r.t. their 20 tensors and the points.  What a
    view-dependent colour head (use_view=True, models/nerf.py:141-153) is built on: it runs as framework ops on
    [feature, encoding(viewdir)], and its gradient w.r.t. the feature re-enters the fused backward through
    anr_mlp_backward_feature.  sigma = -1e5 where valid < 1."""

    @staticmethod
    def forward(ctx, pts, mode_id, *params):
        pts = pts.detach()
        n = pts.shape[0]
        n_pad = max(-(-n // MLPFunction.PAD), 1) * MLPFunction.PAD
        if n_pad != n:
            padded = pts.new_zeros(n_pad, 4)
            padded[:n] = pts
            pts = padded
        out, act = ops.mlp_forward_save(_cached_pack(params, mode_id, False), mode_id, pts)
        ctx.mode_id, ctx.n = mode_id, n
        ctx.save_for_backward(pts, act, *params)
        return out[:n, 3].contiguous(), ops.act_columns(act, 2048, 2304)[:n].float()

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g_sigma, g_feat):
        pts, act, *params = ctx.saved_tensors
        n, n_pad = ctx.n, pts.shape[0]
        weights_generation(params[0], backward=True)
        g_sigma = g_sigma if g_sigma is not None else pts.new_zeros(n)
        g4 = ops.mlp_head_grad(g_sigma, None, None, pts, n, True)                 # (0, 0, 0, d sigma where valid)
        d_feat = pts.new_zeros(n_pad, 256)
        if g_feat is not None:
            d_feat[:n] = g_feat
        dact = ops.mlp_backward_feature(_cached_pack(params, ctx.mode_id, True), ctx.mode_id, g4, d_feat, act)
        flat = ops.mlp_wgrad(ctx.mode_id, act, dact, ops.encode64(pts, act.dtype), g4)
        grads = _split_flat(flat, 20)                                             # the colour head is not this function's
        d_pts = None
        if ctx.needs_input_grad[0]:
            d_enc = ops.mlp_denc(ctx.mode_id, dact, params[0], params[PARAM_KEYS.index("xyz_encoding_5.0.weight")])
            d_pts = ops.encode_backward(pts, d_enc)[:n]
        return (d_pts, None, *[grads[k].to(params[i].dtype) if (grads[k] is not None and ctx.needs_input_grad[2 + i]) else None
                                for i, k in enumerate(PARAM_KEYS)])


class GradSink:
    """One flat fp32 gradient buffer for the 24 tensors of a NeRF, in PARAM_KEYS order — the layout anr_mlp_wgrad writes.
    `p.grad` of every parameter is a VIEW into it, and each MLP backward pass adds its weight gradients with one
    accumulating launch (ANR_MLP_FLAG_ACCUMULATE) instead of handing 24 tensors to autograd, which would add them to
    `.grad` one by one (three passes per network and step: render, sigma priors, normals — ~90 launches).
    A bucket of `training.GradientReducer` is exactly this buffer, so the all-reduce sends it as it is.

    `on_complete` is called when the last backward pass that was announced by a forward pass (`announce`) has
    contributed — the point at which the buffer can go on the wire."""

    def __init__(self, net, flat: torch.Tensor = None):
        named = dict(net.named_parameters())
        self.params = [named[k] for k in PARAM_KEYS]
        total = sum(p.numel() for p in self.params)
        self.flat = flat if flat is not None else torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        assert self.flat.numel() == total and self.flat.dtype == torch.float32
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.expected = self.done = 0
        self.on_complete = None

    def usable(self) -> bool:
        """every parameter trains in fp32 and its .grad is still this buffer's view (a caller may have reset it)"""
        return all(p.requires_grad and p.grad is v for p, v in zip(self.params, self.views))

    def begin_step(self, zero: bool = True):
        if zero:
            self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v
        self.expected = self.done = 0

    def announce(self):
        self.expected += 1

    def contributed(self):
        self.done += 1
        if self.done == self.expected and self.on_complete is not None:
            self.on_complete(self)


def _split_flat(flat, upto=len(PARAM_KEYS)):
    grads, o = {}, 0
    for i, (k, shp) in enumerate(zip(PARAM_KEYS, PARAM_SHAPES)):
        cnt = 1
        for d in shp:
            cnt *= d
        grads[k] = flat[o:o + cnt].view(shp) if i < upto else None
        o += cnt
    return grads


class MLPFunction(torch.autograd.Function):
    """out[n,4] = (r,g,b,sigma) (or sigma[n]) = NeRF(pts[n,4]); differentiable w.r.t. the 24 parameter tensors (and the
    points).  Forward, activation gradients and weight gradients are three hand-written HIP kernels
    (anr_mlp_forward_save, anr_mlp_backward, anr_mlp_wgrad), the steps around them one launch each (csrc/train_glue.hip);
    `LIBRARY_GEMMS = True` swaps the backward kernels for the chain of library GEMMs they replaced (a cross-check for the
    tests).  sink: a GradSink of the network (weight gradients accumulate there, autograd gets None) or None."""

    PAD = 64                 # rows are padded to a multiple of this (anr_mlp_wgrad's slab granularity): invalid points, zero gradient
    LIBRARY_GEMMS = False

    @staticmethod
    def forward(ctx, pts, sigma_only, mode_id, only_valid, sink, *params):
        pack = _cached_pack(params, mode_id, False)
        pts = pts.detach()
        n = pts.shape[0]
        ctx.sigma_only = sigma_only
        ctx.mode_id = mode_id
        ctx.n_full = n
        ctx.sink = sink if (sink is not None and not MLPFunction.LIBRARY_GEMMS and sink.usable()) else None
        if ctx.sink is not None:
            ctx.sink.announce()
        if only_valid:
            # samples outside dis_threshold have sigma = -1e5 and composite weight exactly 0: neither their outputs nor
            # their (exactly zero) gradients are needed.  Forward, saved activations and backward run on the rest, listed
            # in sample order (the row order fixes the order of every split-K sum downstream: same bits on every run).
            # How many there are stays ON THE DEVICE (count[0]; count[1] = padded to 64 rows): every buffer is sized for all n
            # samples — rows that are never written cost address space only — and the kernels stop at the count.  Reading
            # it back to size the buffers drained the queue once per network pass.
            index, pos, pts_c, count = ops.compact_ordered(pts)
            rows_dev = count[1:2]
            out, act = ops.mlp_forward_save(pack, mode_id, pts_c, sigma_only, count=rows_dev)
            ctx.count = count
            ctx.compacted = True
            ctx.save_for_backward(pts_c, out, act, index, pos, *params)
            return ops.expand_rows(out, pos, -1e5)                    # (0,0,0,-1e5) / -1e5 for the samples not listed
        n_pad = max(-(-n // MLPFunction.PAD), 1) * MLPFunction.PAD
        if n_pad != n:
            pts_c = pts.new_zeros(n_pad, 4)                           # padding rows: valid = 0, zero upstream gradient
            pts_c[:n] = pts
            pts = pts_c
        out, act = ops.mlp_forward_save(pack, mode_id, pts, sigma_only)
        ctx.count = None
        ctx.compacted = False
        ctx.save_for_backward(pts, out, act, *params)
        return out[:n] if n_pad != n else out

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        index = pos = None
        if ctx.compacted:
            pts, out, act, index, pos, *params = ctx.saved_tensors
        else:
            pts, out, act, *params = ctx.saved_tensors
        n = pts.shape[0]
        count = ctx.count                                           # compacted: (rows, rows padded to 64) on the device
        rows_dev = None if count is None else count[1:2]
        dt = act.dtype                                              # fp32 (parity mode) or bf16 (mixed precision)
        want_pts = ctx.needs_input_grad[0]
        weights_generation(params[0], backward=True)
        # (dL/d rgb x sigmoid', dL/d sigma where the sample is valid) on the compacted + padded rows
        g4 = ops.mlp_head_grad(g, index, None if ctx.sigma_only else out, pts, ctx.n_full if count is None else count, ctx.sigma_only)
        d_enc = None
        grads = {}
        if MLPFunction.LIBRARY_GEMMS:
            # (the cross-check path works on host-sized tensors: it reads the row count back)
            grads, d_enc = _library_backward(ctx, params, pts, act, g4, want_pts, None if count is None else int(count[1].item()))
            if d_enc is not None and d_enc.shape[0] != pts.shape[0]:
                d_enc = torch.cat([d_enc, d_enc.new_zeros(pts.shape[0] - d_enc.shape[0], d_enc.shape[1])])
        else:
            # activation gradients: ONE kernel for the whole chain (csrc/mlp_bwd.hip); weight + bias gradients: split-K MFMA
            # GEMMs between its output columns, the saved activations and the encoding matrix (csrc/mlp_wgrad.hip)
            dact = ops.mlp_backward(_cached_pack(params, ctx.mode_id, True), ctx.mode_id, g4, act, sigma_only=ctx.sigma_only,
                                    count=rows_dev)
            flat = ops.mlp_wgrad(ctx.mode_id, act, dact, ops.encode64(pts, dt, count=rows_dev), g4, sigma_only=ctx.sigma_only,
                                 accumulate_into=None if ctx.sink is None else ctx.sink.flat, count=rows_dev)
            if ctx.sink is None:
                grads = _split_flat(flat, 18 if ctx.sigma_only else len(PARAM_KEYS))
            if want_pts:                                            # pose refinement: through the two encoding inputs
                d_enc = ops.mlp_denc(ctx.mode_id, dact, params[PARAM_KEYS.index("xyz_encoding_1.0.weight")],
                                     params[PARAM_KEYS.index("xyz_encoding_5.0.weight")], count=rows_dev)
        d_pts = None
        if want_pts:                                                # through x -> (x, sin 2^k x, cos 2^k x)
            d_pts = ops.encode_backward(pts, d_enc.contiguous(), count=rows_dev)
            if pos is not None:
                d_pts = ops.expand_rows(d_pts, pos, 0.0)
            elif d_pts.shape[0] != ctx.n_full:
                d_pts = d_pts[:ctx.n_full]
        out_grads = []
        for i, k in enumerate(PARAM_KEYS):
            need = ctx.needs_input_grad[5 + i]
            gk = grads.get(k) if need else None
            out_grads.append(None if gk is None else gk.to(params[i].dtype).reshape(params[i].shape))
        if ctx.sink is not None:
            ctx.sink.contributed()
        return (d_pts, None, None, None, None, *out_grads)


def _library_backward(ctx, params, pts, act, g4, want_pts, rows=None):
    """The backward of the MLP as the chain of library GEMMs the hand-written kernels replaced (dX = dY W, dW = dY^T X,
    mask kernels in between).  Cross-check only (MLPFunction.LIBRARY_GEMMS): the tests hold the kernels to it.
    rows: the compacted list's padded length (the buffers are sized for all samples; rows past it hold nothing)."""
    dt = act.dtype
    n_all = pts.shape[0]
    n = n_all if rows is None else rows
    cols = lambda c0, c1: ops.act_columns(act, c0, c1)[:n]
    pts, g4 = pts[:n], g4[:n]

    class _Lazy(dict):                                              # parameters in the compute dtype, cast on first use
        def __missing__(self, k):
            p = params[PARAM_KEYS.index(k)]
            self[k] = p if p.dtype == dt else p.to(dt)
            return self[k]
    P = _Lazy()
    H = cols(0, 2048).view(n, 8, 256)
    grads = {}

    def wgrad(dy, x):                                               # accumulated and returned in fp32
        return dy.float().t() @ x.float()

    def relu_bwd(dy, h):
        return torch.ops.aten.threshold_backward(dy, h, 0)
    enc = ops.encode(pts, dt)
    d_enc = None
    g_sig = g4[:, 3].to(dt)
    if ctx.sigma_only:
        dh = g_sig[:, None] * P["sigma.weight"]
    else:
        d_rgb = g4[:, :3].to(dt)
        G, F = cols(2304, 2432), cols(2048, 2304)
        grads["rgb.0.weight"] = wgrad(d_rgb, G)
        grads["rgb.0.bias"] = g4[:, :3].sum(0)
        dG = relu_bwd(d_rgb @ P["rgb.0.weight"], G)
        grads["dir_encoding.0.weight"] = wgrad(dG, F)
        grads["dir_encoding.0.bias"] = dG.sum(0, dtype=torch.float32)
        dF = dG @ P["dir_encoding.0.weight"]
        grads["xyz_encoding_final.weight"] = wgrad(dF, H[:, 7])
        grads["xyz_encoding_final.bias"] = dF.sum(0, dtype=torch.float32)
        dh = torch.addmm(g_sig[:, None] * P["sigma.weight"], dF, P["xyz_encoding_final.weight"])
    grads["sigma.weight"] = wgrad(g_sig[:, None], H[:, 7])
    grads["sigma.bias"] = g4[:, 3].sum().reshape(1)
    for l in range(8, 0, -1):
        dpre = relu_bwd(dh, H[:, l - 1])
        inp = enc if l == 1 else torch.cat([enc, H[:, l - 2]], -1) if l == 5 else H[:, l - 2]
        grads[f"xyz_encoding_{l}.0.weight"] = wgrad(dpre, inp)
        grads[f"xyz_encoding_{l}.0.bias"] = dpre.sum(0, dtype=torch.float32)
        W = P[f"xyz_encoding_{l}.0.weight"]
        if l > 1:
            dh = dpre @ (W[:, 63:] if l == 5 else W)
        if want_pts and l in (1, 5):
            t = (dpre @ W[:, :63]).float()
            d_enc = t if d_enc is None else d_enc + t
    return grads, d_enc


class NormalFunction(torch.autograd.Function):
    """normal[n,3] = d alpha / d xyz with alpha = 1 - exp(-delta relu(sigma(xyz))) (models/nerf.py:177-190), and its
    gradient w.r.t. the trunk and sigma weights — the second-order term of the normals regulariser (train.py:288-309).

    Forward mode instead of autograd-of-autograd: the three tangents d/dx, d/dy, d/dz ride through the trunk next to
    their point — quads of columns in the SAME fused kernels as everything else (ANR_MLP_FLAG_TANGENT: tangent columns
    get the derivative of the encoding, no bias, and the ReLU gate of their point; csrc/mlp_core.h).  ReLU has zero
    curvature almost everywhere, so the backward is the plain linear backward of that 4n-column pass: anr_mlp_backward +
    anr_mlp_wgrad with the same flag (gates from the point's column, bias gradients from the point's column only).
    Three launches forward + backward instead of ~80 library launches (first version) or ~1400 (double backward)."""

    @staticmethod
    def forward(ctx, xyz, delta, mode_id, sink, *params):
        n = xyz.shape[0]
        pts4, act, sig = _tangent_forward(xyz, mode_id, params)
        n_pad = sig.shape[0]
        s0 = sig[:, 0]
        scale = torch.where(s0 > 0, delta * torch.exp(-delta * s0), torch.zeros_like(s0))     # d alpha / d sigma
        scale[n:] = 0
        ctx.save_for_backward(pts4, act, sig, scale, *params)
        ctx.delta, ctx.mode_id, ctx.n = delta, mode_id, n
        ctx.sink = _announce(sink)
        return (scale[:, None] * sig[:, 1:4])[:n]

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        pts4, act, sig, scale, *params = ctx.saved_tensors
        n, n_pad = ctx.n, sig.shape[0]
        d_sig = sig.new_zeros(n_pad, 4)
        d_sig[:n, 1:4] = g * scale[:n, None]
        # d scale / d sigma = -delta * scale where sigma > 0
        d_sig[:n, 0] = (g * sig[:n, 1:4]).sum(-1) * (-ctx.delta) * scale[:n]
        return (None, None, None, None, *_tangent_backward(ctx, params, pts4, act, d_sig, 4))


class QuadSigmaFunction(torch.autograd.Function):
    """quads[n_pad,4] = (sigma, d sigma/dx, d sigma/dy, d sigma/dz) at xyz[n,3] (n_pad = n rounded up to 16; the padding rows
    are zero) — the tangent-mode pass NormalFunction is built on, without the d alpha / d sigma factor: the fused loss
    kernels (anr_train_loss*) apply it, the normalisation and the MSE themselves (train.py:288-309)."""

    @staticmethod
    def forward(ctx, xyz, mode_id, sink, *params):
        pts4, act, sig = _tangent_forward(xyz, mode_id, params)
        ctx.save_for_backward(pts4, act, *params)
        ctx.mode_id = mode_id
        ctx.sink = _announce(sink)
        return sig

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        pts4, act, *params = ctx.saved_tensors
        return (None, None, None, *_tangent_backward(ctx, params, pts4, act, g, 3))


def _announce(sink):
    if sink is not None and sink.usable():
        sink.announce()
        return sink
    return None


def _tangent_forward(xyz, mode_id, params):
    n = xyz.shape[0]
    n_pad = -(-n // 16) * 16                                          # 4 n_pad rows: a multiple of 64 (wgrad slabs)
    pts4 = ops.tangent_quads(xyz.detach().float().contiguous(), n_pad)  # quad p = (point, d/dx, d/dy, d/dz)
    out, act = ops.mlp_forward_save(_cached_pack(params, mode_id, False), mode_id, pts4, sigma_only=True, tangent=True)
    return pts4, act, out.view(n_pad, 4)


def _tangent_backward(ctx, params, pts4, act, d_quads, first_param):
    """Parameter gradients of a tangent-mode pass from dL/d quads[n_pad,4] (list in PARAM_KEYS order; None where the
    gradient went to the network's GradSink or is not needed)."""
    g4 = ops.mlp_head_grad(d_quads.reshape(-1), None, None, pts4, pts4.shape[0], True)
    dact = ops.mlp_backward(_cached_pack(params, ctx.mode_id, True), ctx.mode_id, g4, act, sigma_only=True, tangent=True)
    flat = ops.mlp_wgrad(ctx.mode_id, act, dact, ops.encode64(pts4, act.dtype, tangent=True), g4, sigma_only=True, tangent=True,
                         accumulate_into=None if ctx.sink is None else ctx.sink.flat)
    if ctx.sink is not None:
        ctx.sink.contributed()
        return [None] * len(PARAM_KEYS)
    grads = _split_flat(flat, 18)                                     # the colour head takes no part in sigma
    return [grads[k].to(params[i].dtype) if (grads[k] is not None and ctx.needs_input_grad[first_param + i]) else None
            for i, k in enumerate(PARAM_KEYS)]


class TrainLossFunction(torch.autograd.Function):
    """total, details[10] = every loss term of train.py:228-309 and the weighted total (anr_train_loss), differentiable
    w.r.t. the rendered colours / opacities, the prior sigmas and the tangent quads (anr_train_loss_backward): two
    launches instead of ~180 framework ones."""
    KEYS = ("rgb", "acc", "rgb_fine", "acc_fine", "s", "s_fine", "quads", "quads_fine")

    @staticmethod
    def forward(ctx, consts, target_rgb, target_alpha, *inputs):
        tensors = dict(zip(TrainLossFunction.KEYS, (None if t is None else t.detach() for t in inputs)))
        tensors["target_rgb"], tensors["target_alpha"] = target_rgb, target_alpha
        vals = ops.train_loss(tensors, consts)
        ctx.consts = consts
        ctx.present = [t is not None for t in inputs]
        ctx.save_for_backward(target_rgb, target_alpha, *[t for t in inputs if t is not None])
        details = vals[:10]
        ctx.mark_non_differentiable(details)
        return vals[10].clone(), details

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g_total, _):
        target_rgb, target_alpha, *rest = ctx.saved_tensors
        it = iter(rest)
        tensors = {k: (next(it) if here else None) for k, here in zip(TrainLossFunction.KEYS, ctx.present)}
        tensors["target_rgb"], tensors["target_alpha"] = target_rgb, target_alpha
        want = {k: ctx.needs_input_grad[3 + i] for i, k in enumerate(TrainLossFunction.KEYS)}
        grads = ops.train_loss_backward(tensors, ctx.consts, g_total.float(), want)
        return (None, None, None, *[grads.get(k) for k in TrainLossFunction.KEYS])


class CoarseDepthFunction(torch.autograd.Function):
    """z[bs,R,K] = sample_coarse(rays, steps, t_rand) (models/volume_rendering.py:29-56), differentiable w.r.t. near'/far'
    (columns 6, 7 of the rays in the body frame: they move with the root transform under pose refinement)."""

    @staticmethod
    def forward(ctx, rays, steps, t_rand):
        ctx.save_for_backward(steps, t_rand if t_rand is not None else steps.new_empty(0))
        ctx.shape = rays.shape
        return ops.sample_coarse(rays.detach(), steps, t_rand).view(*rays.shape[:-1], steps.numel())

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        steps, t_rand = ctx.saved_tensors
        d = ops.sample_coarse_backward(g, steps, t_rand if t_rand.numel() else None)
        if ctx.shape[-1] != 8:
            full = d.new_zeros(*ctx.shape)
            full.view(-1, ctx.shape[-1])[:, :8] = d
            return full, None, None
        return d.view(ctx.shape), None, None


class FineMergeFunction(torch.autograd.Function):
    """z_sorted[R,Kc+Kf] = sort(cat(z_coarse, inverse-CDF samples)) (models/volume_rendering.py:59-97,199-207); the fine
    depths are detached in the reference, the coarse ones pass their gradient through the sort's permutation."""

    @staticmethod
    def forward(ctx, z_coarse, weights, u, stash=None):
        zs, perm = ops.sample_fine_merge(z_coarse.detach(), weights, u, want_perm=True)
        if stash is not None:                                   # the fine pass's warp copies the coarse samples' rows by it
            stash["perm"] = perm.to(torch.uint8)
        ctx.save_for_backward(perm)
        ctx.Kc = z_coarse.shape[-1]
        return zs

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        (perm,) = ctx.saved_tensors
        return ops.merge_backward(g, perm, ctx.Kc), None, None, None


class CompositeFunction(torch.autograd.Function):
    """(weights, rgb, depth, acc) = composite(rgbs[R,K,4], z[R,K], rays[R,>=8]); differentiable w.r.t. rgbs."""

    @staticmethod
    def forward(ctx, rgbs, z, rays, noise, white_bkgd):
        w, rgb, depth, acc = ops.composite(rgbs, z, rays, white_bkgd, noise=noise, want_weights=True)
        ctx.save_for_backward(rgbs, z, rays, noise if noise is not None else torch.empty(0, device=z.device))
        ctx.white = white_bkgd
        ctx.mark_non_differentiable(w)
        return w, rgb, depth, acc

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g_w, g_rgb, g_depth, g_acc):
        rgbs, z, rays, noise = ctx.saved_tensors
        R = z.shape[0]
        flat = lambda t: None if t is None else t.reshape(R, -1)               # None: an output the loss does not use
        want_geo = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]          # pose refinement: dL/dz, dL/dfar'
        res = ops.composite_backward(rgbs, z, rays, ctx.white, flat(g_rgb), flat(g_depth), flat(g_acc),
                                     noise=noise if noise.numel() else None, want_dz=want_geo)
        if not want_geo:
            return res, None, None, None, None
        d, dz, dfar = res
        d_rays = torch.zeros_like(rays)
        d_rays[:, 7] = dfar
        return d, dz, d_rays, None, None


class WarpFunction(torch.autograd.Function):
    """pts[bs, R*K, 4] = warp(x = o' + z d') (models/anim_nerf.py:153-192); differentiable w.r.t. the rays in the body
    frame, the depths and the per-vertex observation->canonical transforms.  Neighbour ids and blend weights are
    constants of the backward pass (KNN is `no_grad` in the reference)."""

    @staticmethod
    def forward(ctx, rays, z, o2c, index, lbs_w, thr, skip_far, reuse=None, keep=None):
        # reuse = (pts, nbr_idx, nbr_w, perm) of an earlier call on a subset of these samples (the coarse pass of the step):
        # their rows are copied, not searched again; keep: a dict this call leaves its own (pts, nbr_idx, nbr_w) in
        if reuse is not None and reuse[1] is not None:
            reuse = (reuse[0], None, reuse[3], reuse[1], reuse[2])
        else:
            reuse = None
        pts, nidx, nw = ops.warp_points(index, o2c.detach(), lbs_w, thr, rays=rays.detach(), z=z.detach(),
                                        skip_far=skip_far, neighbours=True, reuse=reuse)
        if keep is not None:
            keep["train"] = (pts, nidx, nw)
        ctx.save_for_backward(rays, z, o2c, nidx, nw)
        return pts

    @staticmethod
    @torch.no_grad()
    def backward(ctx, d_pts):
        rays, z, o2c, nidx, nw = ctx.saved_tensors
        d_o2c, d_rays, d_z = ops.warp_backward(d_pts, rays, z, o2c, nidx, nw)
        if rays.shape[-1] != 8:
            pad = torch.zeros_like(rays)
            pad[..., :8] = d_rays
            d_rays = pad
        return d_rays, d_z, d_o2c, None, None, None, None, None, None


class FrameChainFunction(torch.autograd.Function):
    """Attaches the per-frame chain's backward to the two values of it the renderer consumes, both already produced by the
    forward kernels (anr_smpl_forward / anr_to_root_frame / anr_rays_to_body / anr_ober2cano from the SMPL parameters):
    the rays in the body frame [bs,R,8] and ober2cano[bs,V,4,4] (either may be None).  Backward = ONE anr_frame_backward
    launch for both gradients (forward-mode tangents per parameter, csrc/frame_bwd.hip) instead of torch autograd over
    ~480 tensor ops (smplx/lbs.py:152-251, models/anim_nerf.py:128-151).
    packed = (betas[bs,10], pose[bs,72], transl[bs,3]): the parameter values as the kernels take them."""

    @staticmethod
    def forward(ctx, betas, global_orient, body_pose, transl, rays_body, o2c, consts, rays_world, packed):
        ctx.consts = consts
        ctx.shared = (betas.shape[0] != packed[0].shape[0], transl.shape[0] != packed[2].shape[0])
        ctx.save_for_backward(*packed, rays_world if rays_world is not None else packed[0].new_empty(0))
        return (None if rays_body is None else rays_body.detach(), None if o2c is None else o2c.detach())

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g_rays, g_o2c):
        betas, pose, transl, rays_world = ctx.saved_tensors
        if g_rays is None and g_o2c is None:
            return (None,) * 9
        c = ctx.consts
        grads = ops.frame_backward(betas, pose, transl, c["J0"], c["JS"], c["parents"], c["lbs_weights"], c["shapedirs"],
                                   c["posedirs"], c["T_template"], rays_world=rays_world if g_rays is not None else None,
                                   d_o2c=g_o2c, d_rays=g_rays, vertex_joint_mask=c.get("vjmask"))
        d_betas = grads[:, :10]
        if ctx.shared[0]:
            d_betas = d_betas.sum(0, keepdim=True)
        d_transl = grads[:, 82:85]
        if ctx.shared[1]:
            d_transl = d_transl.sum(0, keepdim=True)
        need = ctx.needs_input_grad
        return (d_betas if need[0] else None, grads[:, 10:13] if need[1] else None, grads[:, 13:82] if need[2] else None,
                d_transl if need[3] else None, None, None, None, None, None)
