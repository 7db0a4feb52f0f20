"""Training step of the rendering path (train.py:102-348 of the reference, minus Lightning).

What is here: the per-step forward (`system_forward`), the reference's photometric / alpha / foreground / background
losses (train.py:228-286), Adam + the polynomial schedule (utils/__init__.py:33-58), and the data-parallel gradient
reduction that replaces Lightning's `strategy='dp'` (config.py:77): one process per GPU, one all-reduce of the
flattened gradient per step (RCCL over xGMI; 2 x 592,388 fp32 = 4.7 MB).

Pose refinement (`optim_body_params`, train.py:141-144,221-222): pass a `BodyModelParams` table; its rows are looked
up per frame, gradients reach them through the differentiable warp (autograd.WarpFunction) and per-frame chain, and
they join the optimiser at half the learning rate and the same gradient all-reduce.

The normals regulariser (train.py:288-309) differentiates d alpha/d xyz a second time: the three directional derivatives
ride through the fused forward / activation-gradient / weight-gradient kernels as forward-mode tangent columns
(`autograd.NormalFunction` / `QuadSigmaFunction`, ANR_MLP_FLAG_TANGENT), and the loss kernels (`anr_train_loss*`) apply
the normalisation and the MSE.
"""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .render import system_forward


@dataclass
class TrainHParams:
    """The fields of cfg.train / cfg the step consumes (config.py:7-101, configs/people_snapshot/*.yaml)."""
    n_samples: int = 64
    n_importance: int = 32
    share_fine: bool = False
    use_unpose: bool = True
    chunk: int = 2048
    lr: float = 5e-4
    lambda_alphas: float = 0.1
    lambda_foreground: float = 0.01
    lambda_background: float = 0.01
    lambda_normals: float = 0.01         # config.py:60; second-order term (forward-mode tangents through the fused kernels)
    epsilon: float = 0.01                # config.py:63
    dis_threshold: float = 0.2
    max_epochs: int = 30
    poly_exp: float = 0.9
    fused_losses: bool = True            # all loss terms in one launch (anr_train_loss) instead of ~100 framework ops


class BodyModelParams(nn.Module):
    """Per-frame learnable SMPL parameters as embedding tables (models/body_model_params.py:5-68, model_type 'smpl'):
    betas shared by all frames (1 x 10), global_orient / transl / body_pose one row per frame.  Same parameter
    names as the reference, so its checkpoints load."""
    DIMS = {"betas": 10, "global_orient": 3, "transl": 3, "body_pose": 69}

    def __init__(self, num_frames, model_type="smpl"):
        super().__init__()
        if model_type != "smpl":
            raise NotImplementedError("only model_type 'smpl' is used by the shipped configs")
        self.num_frames, self.model_type = num_frames, model_type
        self.param_names = list(self.DIMS)
        for name, dim in self.DIMS.items():
            emb = nn.Embedding(1 if name == "betas" else num_frames, dim)
            emb.weight.data.zero_()
            emb.weight.requires_grad = False
            setattr(self, name, emb)

    def init_parameters(self, param_name, data, requires_grad=False):
        if param_name == "betas":
            data = torch.mean(data, dim=0, keepdim=True)
        emb = getattr(self, param_name)
        emb.weight.data = data[..., :self.DIMS[param_name]].to(emb.weight.device)
        emb.weight.requires_grad = requires_grad

    def set_requires_grad(self, param_name, requires_grad=True):
        getattr(self, param_name).weight.requires_grad = requires_grad

    def forward(self, frame_ids):
        return {n: getattr(self, n)(torch.zeros_like(frame_ids) if n == "betas" else frame_ids) for n in self.param_names}


def compute_loss(anim_nerf, hp: TrainHParams, rgbs, alphas, results, fg_points=None, bg_points=None):
    """train.py:228-286 (rgb MSE, alpha L1, foreground / background sigma priors), coarse and fine."""
    fine_net = hp.n_importance > 0 and not hp.share_fine       # a coarse-only model has no `nerf_fine` to probe
    if (hp.fused_losses and rgbs.is_cuda and hasattr(anim_nerf, "_net") and torch.is_grad_enabled()
            and all(anim_nerf._net(f)._hip_supported() for f in ((False, True) if fine_net else (False,)))):
        return _compute_loss_fused(anim_nerf, hp, rgbs, alphas, results, fg_points, bg_points)
    details: Dict[str, torch.Tensor] = {}
    fine = hp.n_importance > 0 and not hp.share_fine
    loss = details.setdefault("loss_rgb", F.mse_loss(results["rgbs"], rgbs))
    if fine:
        details["loss_rgb_fine"] = F.mse_loss(results["rgbs_fine"], rgbs)
        loss = loss + details["loss_rgb_fine"]
    details["loss_alphas"] = F.l1_loss(results["alphas"], alphas)
    loss = loss + hp.lambda_alphas * details["loss_alphas"]
    if fine:
        details["loss_alphas_fine"] = F.l1_loss(results["alphas_fine"], alphas)
        loss = loss + hp.lambda_alphas * details["loss_alphas_fine"]
    k = -2.0 / hp.n_samples
    nets = (("", False),) + ((("_fine", True),) if fine else ())
    want_fg = hp.use_unpose and fg_points is not None
    want_bg = hp.use_unpose and bg_points is not None
    if want_fg or want_bg:
        # one sigma-only field query per network for both point sets (the reference issues them one by one,
        # train.py:262-286; the per-point results are the same, the backward is one pass instead of two)
        both = torch.cat([p for p, w in ((fg_points, want_fg), (bg_points, want_bg)) if w], dim=1)
        n_fg = fg_points.shape[1] if want_fg else 0
        for tag, use_fine in nets:
            s = anim_nerf.query_canonical_space(both, use_fine=use_fine, only_sigma=True)
            if want_fg:
                details["loss_foreground" + tag] = torch.mean(torch.exp(k * torch.relu(s[:, :n_fg])))
                loss = loss + hp.lambda_foreground * details["loss_foreground" + tag]
            if want_bg:
                details["loss_background" + tag] = torch.mean(1 - torch.exp(k * torch.relu(s[:, n_fg:])))
                loss = loss + hp.lambda_background * details["loss_background" + tag]
    if hp.lambda_normals != 0:                                  # train.py:288-309
        pts = anim_nerf.verts_template.detach()
        pts = pts + torch.randn_like(pts) * hp.dis_threshold * 0.5
        nbr = pts + torch.randn_like(pts) * hp.epsilon
        nv = pts.shape[1]
        for tag, use_fine in nets:
            # both point sets through one second-order pass (per-point gradients are independent of the batch)
            nrm = anim_nerf.query_canonical_space(torch.cat([pts, nbr], 1), use_fine=use_fine, only_normal=True)
            nrm = nrm / (torch.norm(nrm, p=2, dim=-1, keepdim=True) + 1e-5)
            details["loss_normals" + tag] = F.mse_loss(nrm[:, :nv], nrm[:, nv:])
            loss = loss + hp.lambda_normals * details["loss_normals" + tag]
    return loss, details


def _compute_loss_fused(anim_nerf, hp, rgbs, alphas, results, fg_points, bg_points):
    """compute_loss with every term, the weighted total and their gradients in two launches (autograd.TrainLossFunction);
    the field queries behind the priors and the normals hand over raw sigmas / tangent quads.  Same random draws, same
    values (tests/test_gpu_training.py::test_fused_losses_equal_the_term_by_term_version)."""
    from . import ops
    from .autograd import TrainLossFunction
    fine = hp.n_importance > 0 and not hp.share_fine
    t = {"rgb": results["rgbs"].reshape(-1, 3), "acc": results["alphas"].reshape(-1)}
    if fine:
        t["rgb_fine"], t["acc_fine"] = results["rgbs_fine"].reshape(-1, 3), results["alphas_fine"].reshape(-1)
    consts = {"R": rgbs.numel() // 3, "k": -2.0 / hp.n_samples, "lambda_alphas": hp.lambda_alphas,
              "lambda_foreground": hp.lambda_foreground, "lambda_background": hp.lambda_background,
              "lambda_normals": hp.lambda_normals}
    tags = [("", False)] + ([("_fine", True)] if fine else [])
    want_fg = hp.use_unpose and fg_points is not None
    want_bg = hp.use_unpose and bg_points is not None
    if want_fg or want_bg:
        both = torch.cat([p for p, w in ((fg_points, want_fg), (bg_points, want_bg)) if w], dim=1)
        consts.update(prior_rows=both.shape[0], n_fg=fg_points.shape[1] if want_fg else 0, n_bg=bg_points.shape[1] if want_bg else 0)
        for tag, use_fine in tags:
            # (their sigma came out of the render pass if the caller attached them there: NeRF.attach_riders)
            rode = anim_nerf._net(use_fine).take_rider_sigma()
            t["s" + tag] = (rode if rode is not None and rode.numel() == both.shape[0] * both.shape[1]
                            else anim_nerf.query_canonical_space(both, use_fine=use_fine, only_sigma=True).reshape(-1))
    if hp.lambda_normals != 0:
        pts = anim_nerf.verts_template.detach()
        pts = pts + torch.randn_like(pts) * hp.dis_threshold * 0.5
        nbr = pts + torch.randn_like(pts) * hp.epsilon
        pair = torch.cat([pts, nbr], 1)
        for tag, use_fine in tags:
            t["quads" + tag] = anim_nerf._net(use_fine).tangent_sigma(pair)
        consts.update(nv=pts.shape[1], normal_sets=pts.shape[0], quad_rows=t["quads"].shape[0], delta=0.02)
    total, vals = TrainLossFunction.apply(consts, rgbs.reshape(-1, 3).contiguous(), alphas.reshape(-1).contiguous(),
                                          *[t.get(k) for k in TrainLossFunction.KEYS])
    present = {"loss_rgb": True, "loss_rgb_fine": fine, "loss_alphas": True, "loss_alphas_fine": fine,
               "loss_foreground": want_fg, "loss_background": want_bg, "loss_foreground_fine": want_fg and fine,
               "loss_background_fine": want_bg and fine, "loss_normals": hp.lambda_normals != 0,
               "loss_normals_fine": hp.lambda_normals != 0 and fine}
    return total, {k: vals[i] for i, k in enumerate(ops.LOSS_NAMES) if present[k]}


def allreduce_gradients(params, world: Optional[int] = None):
    """Average gradients over ranks with ONE collective on a flat buffer (DataParallel's reduce_add, train.py:454-455,
    as an all-reduce).  The buffer covers EVERY parameter of the list, in list order, whether or not this rank produced
    a gradient for it (a rank whose rays all miss the body has none: it contributes zeros and receives the average), so
    all ranks issue the same collective.  No-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    params = list(params)
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch._utils._flatten_dense_tensors(grads)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world or dist.get_world_size())
    for p, g, f in zip(params, grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        if p.grad is None:
            p.grad = f.clone()
        else:
            g.copy_(f)
    return flat.numel()


class GradientReducer:
    """Bucketed gradient averaging overlapped with the backward pass (what Lightning's `strategy='dp'` does with a
    per-step broadcast + reduce_add, config.py:77 / train.py:451-458, as RCCL all-reduces over xGMI).

    * `buckets`: lists of parameters, in the order their gradients complete during backward (fine network first).  Each
      bucket owns one flat fp32 buffer; `p.grad` of its parameters are VIEWS into it, so autograd accumulates straight into
      the send buffer and nothing is copied before or after the collective.
    * a post-accumulate hook per parameter counts arrivals; a full bucket is all-reduced asynchronously (RCCL runs on its
      own stream) while autograd keeps going.  Buckets are always issued in list order, and `finish()` issues whatever is
      left — a rank that produced no gradient at all (every ray missed the body) still joins every collective, with zeros.
    * xGMI is point-to-point: a ring all-reduce of a 2.4 MB bucket is latency-bound (tens of microseconds), so two or three
      buckets are the useful granularity here, not DDP's 25 MB.
    """

    def __init__(self, buckets, world: Optional[int] = None):
        import torch.distributed as dist
        self.dist = dist
        self._world_arg = world
        self.active, self.world = False, 1                   # decided per step in prepare(): the process group may be
        self._probe()                                         # initialised after the Trainer was built
        self.buckets = [list(b) for b in buckets if len(b)]
        self.flat, self.slot = [], {}                         # slot[p] = (bucket index, p's view into the bucket's buffer)
        # the buckets lie back to back in ONE buffer (`whole`; each bucket's start 16-byte aligned): the graphed step reduces all
        # of it with one collective, the eager step's hooks reduce the buckets — views of it — one by one as they complete
        same = len({(b[0].dtype, b[0].device) for b in self.buckets}) <= 1
        sizes = [(sum(p.numel() for p in b) + 3) // 4 * 4 for b in self.buckets]
        self.whole = torch.zeros(sum(sizes), dtype=self.buckets[0][0].dtype, device=self.buckets[0][0].device) if (self.buckets and same) else None
        start = 0
        for bi, b in enumerate(self.buckets):
            n_b = sum(p.numel() for p in b)
            # (.data: the bucket shares the storage, NOT the version counter — the hooks below tell a write to THIS bucket by its
            # counter, and views of one tensor share theirs)
            flat = (self.whole[start:start + n_b].data if self.whole is not None
                    else torch.zeros(n_b, dtype=b[0].dtype, device=b[0].device))
            start += sizes[bi]
            o = 0
            for p in b:
                self.slot[p] = (bi, flat[o:o + p.numel()].view_as(p))
                o += p.numel()
            self.flat.append(flat)
        for p in self.slot:
            p.register_post_accumulate_grad_hook(self._arrived)
        self._pending, self._next, self._handles = [], 0, []
        self.sinks = []

    def _probe(self):
        dist = self.dist
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = self._world_arg or (dist.get_world_size() if self.active else 1)

    def attach_sink(self, net):
        """If the 24 tensors (PARAM_KEYS) of `net` lie back to back in PARAM_KEYS order in one bucket, make that stretch the network's
        GradSink: its MLP backward passes accumulate straight into the send buffer and report completion per network."""
        from .autograd import PARAM_KEYS, GradSink
        named = dict(net.named_parameters())
        params = [named.get(k) for k in PARAM_KEYS]
        if any(p is None or p not in self.slot or not p.requires_grad or p.dtype != torch.float32 for p in params):
            return None
        bi = self.slot[params[0]][0]
        views = [self.slot[p][1] for p in params]
        first = views[0].data_ptr()
        o = 0
        for p, v in zip(params, views):
            if self.slot[p][0] != bi or v.data_ptr() != first + 4 * o:
                return None
            o += p.numel()
        start = (first - self.flat[bi].data_ptr()) // 4
        sink = GradSink(net, self.flat[bi][start:start + o])
        sink.views = views                                    # the very tensors prepare() installs as .grad
        sink.on_complete = lambda s, bi=bi, k=len(params): self._sink_done(bi, k)
        net.grad_sink = sink
        self.sinks.append(sink)
        return sink

    def _sink_done(self, bi, k):
        """the network's last announced backward pass has added its gradients: its k tensors have arrived"""
        if self.active:
            self._take(bi, k)
            self._issue_ready()

    def _take(self, bi, k, who=None):
        if bi < self._next:
            state = [(s.expected, s.done) for s in self.sinks]
            raise RuntimeError(f"a gradient for bucket {bi} arrived after its all-reduce was issued: a backward pass on a "
                               "sink-attached network delivered gradients through autograd after the sink had reported "
                               f"completion (announce() every pass in its forward, or detach the sink); tensor {who}, "
                               f"sinks (expected, done) = {state}")
        self._pending[bi] -= k
        assert self._pending[bi] >= 0, (bi, self._pending)

    def prepare(self, zero: bool = True):
        """Before backward: zero the send buffers and point every p.grad at its slice.  zero=False: the caller clears them
        itself (the explicit step: the library's fill kernel instead of a framework launch)."""
        self._probe()
        self._pending = [len(b) for b in self.buckets]
        self._sink_params = {id(p) for s in self.sinks for p in s.params}
        self._next, self._handles = 0, []
        for flat in self.flat:
            if zero:
                flat.zero_()
        for p, (_, view) in self.slot.items():
            p.grad = view
        for sink in self.sinks:
            sink.begin_step(zero=False)
        # every .grad of a bucket is a view of its flat buffer and shares its version counter: autograd's in-place
        # accumulation raises it, the kernels that add into the buffer through raw pointers (GradSink) do not
        self._ver = [f._version for f in self.flat]

    # deferred: hooks and finish() issue nothing; the caller reduces the buffers itself afterwards (`reduce_flat`).  What a
    # captured forward + backward runs with: a collective cannot be issued from inside the capture on every backend, and
    # two 2.4 MB buckets cost ~0.1 ms against a 5 ms step, so overlapping them with backward buys nothing there.
    deferred = False

    def reduce_flat(self):
        """All-reduce + average the flat buffers now (the deferred mode's counterpart of the hooks + finish()): ONE collective
        over the whole buffer — on xGMI a 2.4 MB ring all-reduce is latency-bound, so one of 4.8 MB costs about what each of two
        did — averaged by the collective itself on RCCL (ReduceOp.AVG: no division launch); gloo sums, then one division."""
        self._probe()
        if not self.active:
            return 0
        avg = self.dist.get_backend() == "nccl" and self._world_arg is None
        for flat in ([self.whole] if self.whole is not None else self.flat):
            self.dist.all_reduce(flat, op=self.dist.ReduceOp.AVG if avg else self.dist.ReduceOp.SUM)
            if not avg:
                flat.div_(self.world)
        return sum(f.numel() for f in self.flat)

    def reduce_flat_async(self, bi):
        """Issue the all-reduce of bucket `bi` now, asynchronously (the backend orders it behind what the current stream has
        issued); -> [(handle, flat, needs_division)].  RCCL averages in the collective itself (ReduceOp.AVG); gloo sums."""
        self._probe()
        if not self.active:
            return []
        flat = self.flat[bi]
        avg = self.dist.get_backend() == "nccl" and self._world_arg is None
        h = self.dist.all_reduce(flat, op=self.dist.ReduceOp.AVG if avg else self.dist.ReduceOp.SUM, async_op=True)
        return [(h, flat, not avg)]

    def finish_async(self, pending):
        for h, flat, divide in pending:
            h.wait()                                          # (the current stream waits for the collective)
            if divide:
                flat.div_(self.world)

    def _issue_ready(self, force=False):
        if self.deferred:
            return
        while self._next < len(self.buckets) and (force or self._pending[self._next] == 0):
            self._handles.append(self.dist.all_reduce(self.flat[self._next], op=self.dist.ReduceOp.SUM, async_op=True))
            self._next += 1

    def _arrived(self, p):
        if not self.active or not self._pending:
            return
        bi, view = self.slot[p]
        if p.grad is not view:                                # a caller reset .grad after prepare(): move the value in
            view.copy_(p.grad)
            p.grad = view
        wrote = self.flat[bi]._version != self._ver[bi]
        self._ver[bi] = self.flat[bi]._version
        if id(p) in self._sink_params:
            # A sink-owned tensor: its arrival is counted ONCE, by the sink's completion (all k tensors together).  Its hook
            # still fires when autograd is done with the tensor — normally with nothing to accumulate (the passes returned
            # None and added through the sink).  If autograd DID write (a pass that ran without the sink: FeatureFunction,
            # a sink found unusable), the bucket must not have left yet.  A network whose passes all bypass the sink is sent
            # by finish().
            if wrote:
                self._take(bi, 0, who=(tuple(p.shape), [i for i, q in enumerate(self.buckets[bi]) if q is p]))
            return
        self._take(bi, 1)
        self._issue_ready()

    def finish(self):
        """After backward: issue what is left (in order), wait, average.  Returns the number of floats reduced."""
        if self.deferred:
            return 0
        if not self.active:
            if self.dist.is_available() and self.dist.is_initialized() and self.dist.get_world_size() > 1:
                raise RuntimeError("the process group came up between prepare() and finish(): gradients of this step "
                                   "were not set up for reduction")
            return 0
        self._issue_ready(force=True)
        for h in self._handles:
            h.wait()
        for flat in self.flat:
            flat.div_(self.world)
        return sum(f.numel() for f in self.flat)


class FlatAdam(torch.optim.Optimizer):
    """`torch.optim.Adam(params, lr, betas, eps)` (train.py:216-226; amsgrad / maximize off, no weight decay) as ONE kernel
    launch per step over every tensor of every group (`anr_adam_step`): the moments live in two flat buffers (`state[p]`
    holds views, so `state_dict()` has torch.optim.Adam's layout), the step counters (one per tensor, as torch keeps them) on the device (a captured step
    replays with the right bias corrections), the learning rates stay host floats in `param_groups` (schedulers work as usual; a
    graph capture bakes them in).  A parameter whose `.grad` is None is skipped, as torch does."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        # (the keys torch.optim.Adam's param_groups carry, with the values this optimiser implements: a state_dict() of
        # GPU training loads into torch.optim.Adam — the CPU / non-contiguous fallback and what the reference uses — and steps)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        ps = [p for g in self.param_groups for p in g["params"]]
        if not ps or len(self.param_groups) > 4:
            raise ValueError("FlatAdam: 1 to 4 parameter groups with at least one tensor")
        if any(not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() for p in ps):
            raise ValueError("FlatAdam: contiguous float32 parameters on the GPU (the CPU path is torch.optim.Adam)")
        if any(g["betas"] != self.param_groups[0]["betas"] or g["eps"] != self.param_groups[0]["eps"] for g in self.param_groups):
            raise ValueError("FlatAdam: one (betas, eps) for all groups")
        dev = ps[0].device
        offs, o = [], 0
        for p in ps:
            offs.append(o)
            o += (p.numel() + 3) // 4 * 4                       # 16-byte aligned slices: the kernel moves float4
        self._m = torch.zeros(o, dtype=torch.float32, device=dev)
        self._v = torch.zeros(o, dtype=torch.float32, device=dev)
        self._steps = torch.zeros(len(ps), dtype=torch.float32, device=dev)   # one counter per tensor, as torch keeps them
        self._active = torch.zeros(len(ps), dtype=torch.float32, device=dev)  # 1 where the tensor has a gradient this step
        self._ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self._index = {p: i for i, p in enumerate(ps)}
        self._slices = {p: (off, p.numel()) for p, off in zip(ps, offs)}
        self._bind_state()
        self._table, self._table_key, self._n_chunks = None, None, 0

    def _bind_state(self):
        for p, (off, n) in self._slices.items():
            self.state[p] = {"step": self._steps[self._index[p]], "exp_avg": self._m[off:off + n].view_as(p),
                             "exp_avg_sq": self._v[off:off + n].view_as(p)}

    def _build_table(self):
        import numpy as np
        from . import _lib
        key = tuple((p.data_ptr(), None if p.grad is None else p.grad.data_ptr()) for p in self._slices)
        if key == self._table_key:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FlatAdam: parameter or gradient storage changed inside a graph capture (run one eager step first)")
        lib = _lib.load()
        chunk = lib.anr_adam_chunk_floats()
        rec = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("count", "<i4"), ("group", "<i4")])
        assert rec.itemsize == lib.anr_adam_chunk_bytes()
        rows = []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.shape != p.shape:
                    raise ValueError("FlatAdam: dense contiguous float32 gradients of the parameter's shape")
                off, n = self._slices[p]
                for c in range(0, n, chunk):
                    rows.append((p.data_ptr() + 4 * c, p.grad.data_ptr() + 4 * c, self._m.data_ptr() + 4 * (off + c),
                                 self._v.data_ptr() + 4 * (off + c), min(chunk, n - c), gi | (self._index[p] << 8)))
        self._n_chunks = len(rows)
        self._active.copy_(torch.tensor([0.0 if p.grad is None else 1.0 for p in self._slices], dtype=torch.float32))
        arr = np.array(rows, dtype=rec) if rows else np.zeros(1, dtype=rec)
        self._table = torch.from_numpy(arr.view(np.uint8).copy()).to(self._m.device)
        self._table_key = key

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._build_table()
        if self._n_chunks:
            from . import ops
            g0 = self.param_groups[0]
            # (the kernel advances the per-tensor step counters itself: no increment launch in front of it)
            ops.adam_step(self._table, self._n_chunks, self._steps, [float(g["lr"]) for g in self.param_groups],
                          g0["betas"][0], g0["betas"][1], g0["eps"], active=self._active, ticket=self._ticket)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)                     # (torch replaces the state tensors by copies of the loaded ones)
        loaded = dict(self.state)
        self._bind_state()
        with torch.no_grad():
            for p, st in loaded.items():
                if p in self._slices and "exp_avg" in st:
                    self.state[p]["exp_avg"].copy_(st["exp_avg"])
                    self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                    self._steps[self._index[p]] = float(st["step"])


class Trainer:
    """optimizer + schedule + step; `step(batch)` mirrors AnimNeRFSystem.training_step (train.py:324-348)."""

    def __init__(self, anim_nerf, volume_renderer, hp: TrainHParams, body_model_params: Optional[BodyModelParams] = None,
                 graph: bool = False, explicit_step: bool = True, static_inputs: bool = False):
        """graph=True: `step_graphed` may capture the whole step (forward, losses, backward, Adam) into ONE HIP graph and
        replay it — every launch of the step leaves the host as one (the step holds no device -> host read: row counts stay
        on the device; FlatAdam keeps its step counter there too).  Opt-in.  `with trainer.loop():` runs the loop on the
        Trainer's stream and saves the two stream fences per step; it is an optimisation, not a requirement: the GPU memory
        fault long replayed runs used to end in was a hipMemsetAsync NODE of the captured graph going stale (ROCm 7.2;
        DESIGN.md section 4.4, tools/exp/graph_hazard_torch_only.py) and the library issues no memset any more.
        explicit_step=True (default): a step of the shipped configuration runs as one explicit sequence of library launches
        (fused_step.ExplicitTrainStep: forward, losses, backward, every gradient accumulated in place — no autograd graph, no
        framework kernel between the first launch and the last); any other configuration, or a caller that patches torch's
        random functions to inject draws, goes through the autograd Functions (autograd.py).  False: always autograd.
        static_inputs=True (opt-in): a replayed step does not copy a batch tensor that is the very object, at the very version
        counter, it copied last time — for a loader that REUSES its batch buffers and writes them through torch (a writer
        that does not bump the version counter — `x.data.copy_()`, DLPack views, raw-pointer kernels — would train on a stale
        batch).  Default: every batch tensor is copied every step, all of them in one launch (anr_copy_segments)."""
        self.static_inputs = bool(static_inputs)
        self.model, self.renderer, self.hp = anim_nerf, volume_renderer, hp
        self.body_model_params = body_model_params
        self.graph_enabled = bool(graph)
        self._graph = None                                    # (signature, CUDAGraph, static inputs, static outputs, pins)
        self._leaf_seen = []
        self._graph_warm = 0
        self._graph_split = False                             # the graph ends with backward (more than one rank)
        self._graph_second = None                             # ... and is cut in two at the fine network's completed bucket
        for name, p in anim_nerf.named_parameters():          # SMPL member params are unused by the forward
            if name.startswith("body_model."):
                p.requires_grad_(False)
        groups = [{"params": [p for p in anim_nerf.parameters() if p.requires_grad], "lr": hp.lr}]
        if body_model_params is not None:                     # train.py:221-222: half the learning rate
            bp = [p for p in body_model_params.parameters() if p.requires_grad]
            if bp:
                groups.append({"params": bp, "lr": hp.lr * 0.5})
        self.params = [p for g in groups for p in g["params"]]
        # (one fused update kernel per parameter group on the GPU instead of seven list kernels)
        on_gpu = bool(self.params and self.params[0].is_cuda)
        if on_gpu and all(p.dtype == torch.float32 and p.is_contiguous() for p in self.params):
            # one launch for every tensor of both groups, step counter on the device (capturable).  The learning rates stay host
            # floats — a capture bakes them in, and step_graphed captures again when the scheduler has moved them: once per epoch.
            self.optimizer = FlatAdam(groups, eps=1e-8)
        else:
            self.optimizer = torch.optim.Adam(groups, eps=1e-8, weight_decay=0, fused=on_gpu)
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(
            self.optimizer, lambda epoch: (1 - epoch / hp.max_epochs) ** hp.poly_exp)
        # gradient buckets in the order backward completes them: fine network, coarse network, SMPL parameter rows
        # (inside a network: the order anr_mlp_wgrad writes, so that the bucket is the kernel's output buffer)
        from .autograd import PARAM_KEYS

        def ordered(net):
            named = dict(net.named_parameters())
            head = [named[k] for k in PARAM_KEYS if k in named]
            ids = {id(p) for p in head}
            return [p for p in head + [p for p in net.parameters() if id(p) not in ids] if p.requires_grad]
        nets = [getattr(anim_nerf, "nerf_fine", anim_nerf.nerf)]
        fine = ordered(nets[0])
        fine_ids = {id(p) for p in fine}
        coarse = [p for p in ordered(anim_nerf.nerf) if id(p) not in fine_ids]
        if coarse:
            nets.append(anim_nerf.nerf)
        seen = fine_ids | {id(p) for p in coarse}
        rest = [p for p in self.params if id(p) not in seen]
        # A capture must never touch the legacy default stream, and autograd synchronises the stream an AccumulateGrad node was
        # CREATED under with the stream that produces its gradient.  A graphed Trainer therefore owns a stream: the hooks below
        # (which create those nodes), every eager step, the capture and the replays all run on it, fenced against the caller's
        # current stream on both sides.
        self._stream = torch.cuda.Stream(self.params[0].device) if (self.graph_enabled and on_gpu) else None
        self.explicit = None                                  # fused_step.ExplicitTrainStep (GPU, the shipped configuration)
        # (construction enqueues nothing the caller's stream must see: no fence against it here)
        with (torch.cuda.stream(self._stream) if self._stream is not None else contextlib.nullcontext()):
            self.reducer = GradientReducer([fine, coarse + rest])
            if on_gpu:
                for net in nets:
                    self.reducer.attach_sink(net)
                if explicit_step:
                    from .fused_step import ExplicitTrainStep
                    self.explicit = ExplicitTrainStep(self)

    @property
    def stream(self):
        """The stream a graphed Trainer works on (None otherwise)."""
        return self._stream

    @contextlib.contextmanager
    def loop(self):
        """Run a training loop on the Trainer's stream: `with trainer.loop(): for batch in ...: trainer.step_graphed(...)`.
        Everything inside — the steps, progress reads (`loss.item()`), validation renders — then shares ONE stream with the
        replays, and `step_graphed` issues no fence against the caller's stream (two event records + waits per step otherwise).
        No-op for a Trainer without graph=True."""
        with self._own_stream():
            yield self

    @contextlib.contextmanager
    def _own_stream(self):
        cur = torch.cuda.current_stream(self._stream.device) if self._stream is not None else None
        if cur is None or cur == self._stream:
            yield
            return
        self._stream.wait_stream(cur)
        try:
            with torch.cuda.stream(self._stream):
                yield
        finally:
            cur.wait_stream(self._stream)

    def _attach_prior_points(self, fg_points, bg_points):
        """The prior points' sigma comes out of the render passes (NeRF.attach_riders) when the fused losses will ask for it."""
        import os
        hp, m = self.hp, self.model
        pts = [p for p in (fg_points, bg_points) if p is not None]
        fine = hp.n_importance > 0 and not hp.share_fine
        nets = [m._net(f) for f in ((False, True) if fine else (False,))] if hasattr(m, "_net") else []
        on = (hp.fused_losses and hp.use_unpose and pts and pts[0].is_cuda and nets and all(n._hip_supported() for n in nets)
              and not os.environ.get("ANR_TRAIN_SEPARATE_PRIOR_QUERY"))
        for n in nets:
            n.attach_riders(torch.cat(pts, dim=1) if on else None)

    def begin_step(self):
        """Zero the flat gradient buffers and point every p.grad at its slice (instead of optimizer.zero_grad)."""
        self.reducer.prepare()

    # eager steps before a capture: lazy initialisation (library handles, allocator pools, autograd's threads) must not
    # happen inside a capture
    GRAPH_WARM_STEPS = 3

    def step_graphed(self, rays, rgbs, alphas, body_model_params, body_model_params_template, fg_points=None, bg_points=None,
                     perturb=1.0, frame_idx=None):
        """`step` with the same arguments and results, replayed from a HIP graph once the shapes have been seen
        GRAPH_WARM_STEPS times (those steps, and every step of a Trainer built without graph=True, run eagerly).  In a
        process group with more than one rank the graph holds forward + backward; the all-reduce of the two flat gradient
        buffers and Adam follow each replay.  Inputs are copied into the graph's own buffers, the loss and the details
        come back as fresh tensors; random draws (stratified offsets, sigma noise, loss points) advance per replay through
        the generator state torch registers with the graph."""
        eager = not self.graph_enabled or not rays.is_cuda
        args = {"rays": rays, "rgbs": rgbs, "alphas": alphas, "bmp": body_model_params, "templ": body_model_params_template,
                "fg": fg_points, "bg": bg_points, "frame_idx": frame_idx}

        def flat(v, prefix):
            if isinstance(v, dict):
                return [x for k in sorted(v) for x in flat(v[k], f"{prefix}.{k}")]
            return [(prefix, v)] if torch.is_tensor(v) else []
        leaves = [x for k in sorted(args) for x in flat(args[k], k)]
        with self._own_stream():
            return self._step_graphed(args, leaves, perturb, eager)

    def _step_graphed(self, args, leaves, perturb, eager):
        dist = torch.distributed
        # (ANR_GRAPH_FORCE_SPLIT=1: the more-than-one-rank structure — backward cut in two graphs, Adam behind them — on one rank,
        # to time what the cut costs: tools/time_train_cfg4.py)
        split = (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) or bool(os.environ.get("ANR_GRAPH_FORCE_SPLIT"))
        shapes = (float(perturb), bool(split), tuple((n, tuple(t.shape), t.dtype) for n, t in leaves))
        # host values a capture freezes (none when the optimiser step stays outside the graph)
        baked = () if split else tuple(float(g["lr"]) for g in self.optimizer.param_groups)
        # Frozen networks (`_refine`): their weight packs are made ONCE, found in the cache by the capture and read by every
        # replay from the address the graph pinned.  A checkpoint loaded after the first capture (load_state_dict, p.copy_)
        # bumps the tensors' version counters: the signature carries them, so the step is captured again — with fresh packs —
        # instead of replaying on the old weights.  (A writer that does not bump the counter — `p.data.copy_()` — is invisible
        # to every cache keyed by it; torch.autograd.graph.increment_version is the remedy, INTEGRATION.md.)
        frozen_ver = ()
        if self.explicit is not None and self.explicit.frozen_networks():
            m = self.model
            frozen_ver = tuple((id(p), p._version) for net in (m.nerf, getattr(m, "nerf_fine", None)) if net is not None
                               for p in net.parameters())
        sig = (shapes, baked, frozen_ver)
        if not eager and (self._graph is None or self._graph[0] != sig):
            if self._graph is not None and self._graph[0][0] != shapes or (self._graph is None and self._graph_warm < self.GRAPH_WARM_STEPS):
                # (other shapes — the short last batch of an epoch — are not captured: one graph per Trainer, the step stays
                # correct through the eager path)
                self._graph_warm += 1
                eager = True
            else:
                # first capture, or the scheduler moved the learning rates: capture again
                try:
                    self._capture(sig, args, leaves, perturb)
                except Exception as exc:                        # a capture that fails must not cost the step: eager from here on
                    import warnings
                    from .autograd import bump_generation
                    warnings.warn(f"Trainer: graph capture failed ({type(exc).__name__}: {exc}); continuing with eager steps")
                    self._graph, self.graph_enabled, eager = None, False, True
                    self.reducer.deferred = False
                    bump_generation(self.params)                # nothing cached during the capture survives it
        if eager:
            return self.step(args["rays"], args["rgbs"], args["alphas"], args["bmp"], args["templ"], args["fg"], args["bg"],
                             perturb=perturb, frame_idx=args["frame_idx"])
        # inputs -> the graph's own buffers, all in one launch.  The batch is copied every step unless static_inputs=True (then a
        # tensor that is the very object, at the very version, copied last time is skipped); the TEMPLATE pose — one dict for a
        # whole run in the reference's loader (datasets/anim_nerf_dataset.py:278), ten scalars' worth of tensors — is always
        # compared by object + version: when it is new its body state, which lives OUTSIDE the graph (computed once per run,
        # models/anim_nerf.py:108-126), is recomputed by a few eager launches INTO the tensors the graph reads — no re-capture,
        # whatever the loader does (it used to re-capture, with a device synchronise, on every fresh template object).
        from . import ops
        templ_changed, todo = False, []
        for i, ((name, src), dst) in enumerate(zip(leaves, self._graph[2])):
            seen = self._leaf_seen[i]
            is_templ = name.startswith("templ")
            if dst.data_ptr() == src.data_ptr():
                continue
            if (is_templ or self.static_inputs) and seen is not None and seen[0] is src and seen[1] == src._version:
                continue
            if src.is_contiguous() and src.dtype == dst.dtype and src.device == dst.device:
                todo.append((dst, src))
            else:
                dst.copy_(src, non_blocking=True)
            self._leaf_seen[i] = (src, src._version)
            templ_changed |= is_templ
        if todo:
            ops.copy_segments(todo)
        if templ_changed:
            self._refresh_template_state()
        _, graph, static_leaves, outs, _pins, _st = self._graph
        graph.replay()
        if self._graph_split:                                  # more than one rank: the graph ends with backward
            if self._graph_second is not None:
                # the fine network's bucket is complete: its all-reduce (the backend's own stream, behind what this stream has
                # issued so far) runs while the second graph replays
                pending = self.reducer.reduce_flat_async(0)
                self._graph_second.replay()
                pending += self.reducer.reduce_flat_async(1)
                self.reducer.finish_async(pending)
            else:
                self.reducer.reduce_flat()
            self.optimizer.step()
        from .autograd import bump_generation
        bump_generation(self.params)                          # the packs cached under the old generation belong to the graph
        loss, details = outs
        # fresh tensors for the caller: outputs that are views of one buffer (the loss kernel's 12 values) share ONE copy
        copies = {}

        def fresh(v):
            if not torch.is_tensor(v):
                return v
            base = v._base if v._base is not None else v
            c = copies.get(id(base))
            if c is None:
                c = copies[id(base)] = base.clone()
            return c if v._base is None else torch.as_strided(c, v.size(), v.stride(), v.storage_offset() - base.storage_offset())
        return fresh(loss), {k: fresh(v) for k, v in details.items()}

    def _capture(self, sig, args, leaves, perturb):
        def rebuild(v, it):
            if isinstance(v, dict):
                return {k: rebuild(v[k], it) for k in sorted(v)}
            return next(it) if torch.is_tensor(v) else v
        static_leaves = [t.clone() for _, t in leaves]
        self._leaf_seen = [(t, t._version) for _, t in leaves]
        it = iter(static_leaves)
        st = {k: rebuild(args[k], it) for k in sorted(args)}
        if st["templ"] is not None and hasattr(self.model, "_same_template"):
            with torch.no_grad():                              # the template's body state: outside the graph (see the replay)
                if not self.model._same_template(st["templ"]):
                    self.model._set_template(st["templ"])
        torch.cuda.synchronize()
        self._graph = None                                    # (a previous capture's pool goes back to the allocator first)
        # More than one rank: the graph holds forward + backward into the flat buffers; the all-reduce of those buffers and
        # Adam follow each replay eagerly (3 launches + 2 collectives instead of ~190 launches).
        split = sig[0][1]
        graph = torch.cuda.CUDAGraph()
        self.reducer.deferred = split
        # More than one rank: ONE graph (forward + backward into the flat gradient buffer), then ONE all-reduce of the whole
        # buffer and Adam.  ANR_GRAPH_SPLIT=1 (opt-in, measured and not the default): the step captured as TWO graphs, cut where
        # the fine network's flat gradient is complete (ExplicitTrainStep.run's `at_split`), so that a replayed step issues the
        # all-reduce of that bucket while the second graph — the coarse pass's backward, the pose chain — replays.  A capture
        # can only end with every branch joined, so the cut makes the fine pass's weight gradients and the whole normals branch
        # finish before the coarse backward starts: +83 us on the 1.34 ms step of the reference's per-rank batch on one GPU
        # (gpurun_out/r05/step_timings.txt), more than one 2.4 MB all-reduce over xGMI costs.  (An event-record NODE in the one
        # graph would let a side stream start the collective mid-replay without a cut: torch refuses external events on ROCm
        # — tools/exp/graph_external_event.py.)  The second capture shares the first one's memory pool and the two are always
        # replayed in this order.
        two = (split and self.explicit is not None and len(self.reducer.flat) == 2 and bool(os.environ.get("ANR_GRAPH_SPLIT"))
               and self.explicit.supported(st["rays"], st["bmp"], st["frame_idx"], st["fg"], st["bg"]) and not self.explicit.frozen_networks())
        second = torch.cuda.CUDAGraph() if two else None
        cut = {"done": False}

        def at_split():
            graph.capture_end()
            second.capture_begin(pool=graph.pool())
            cut["done"] = True
        try:
            if two:
                import gc
                torch.cuda.synchronize()
                gc.collect()
                torch.cuda.empty_cache()
                with torch.cuda.stream(self._stream):
                    graph.capture_begin()
                    try:
                        loss, details = self._step_body(st["rays"], st["rgbs"], st["alphas"], st["bmp"], st["templ"], st["fg"], st["bg"],
                                                        perturb, st["frame_idx"], apply=False, at_split=at_split)
                    finally:
                        (second if cut["done"] else graph).capture_end()
                if not cut["done"]:
                    second = None
            else:
                with torch.cuda.graph(graph, stream=self._stream):
                    loss, details = self._step_body(st["rays"], st["rgbs"], st["alphas"], st["bmp"], st["templ"], st["fg"], st["bg"],
                                                    perturb, st["frame_idx"], apply=not split)
        finally:
            self.reducer.deferred = False
        self._graph_split = split
        self._graph_second = second
        # Every address the capture baked in that is NOT in the graph's private pool must outlive the graph: module-level
        # scratch that is REPLACED when a later eager call needs more (ops._WGRAD_WS, ops._LOSS_WS) and FlatAdam's chunk table
        # (rebuilt when a .grad pointer moves).  The graph pins what it saw; a replacement allocates next to it.
        from . import autograd, ops
        pins = [list(ops._WGRAD_WS.values()), list(ops._LOSS_WS.values()), list(ops._COMPACT_STATE.values()), getattr(self.optimizer, "_table", None),
                # weight packs the capture found in the cache instead of making them (frozen networks: made once per run)
                [hit[1] for hit in autograd._PACKS.values()],
                [p.grad for p in self.params],
                # the template pose's body state, computed outside the graph (an eager call with another template replaces
                # the model's attributes, not these tensors)
                [getattr(self.model, n, None) for n in self._TEMPLATE_STATE]]
        self._graph = (sig, graph, static_leaves, (loss, details), pins, st)

    _TEMPLATE_STATE = ("verts_template", "joints_template", "verts_transform_template", "joints_transform_template",
                       "shape_offsets_template", "pose_offsets_template")

    def _refresh_template_state(self):
        """New template values in the graph's static template leaves: recompute the template pose's body state eagerly and
        write it INTO the tensors the captured step reads (the graph pins them), instead of capturing again."""
        st, m = self._graph[5], self.model
        old = [getattr(m, n) for n in self._TEMPLATE_STATE]
        with torch.no_grad():
            m._set_template(st["templ"])
            for n, o in zip(self._TEMPLATE_STATE, old):
                o.copy_(getattr(m, n))
                setattr(m, n, o)
            m._same_template(st["templ"])                       # (the cache now describes the static leaves at their new version)

    def state_dict(self):
        """Everything a resumed run needs besides the model's own state_dict: the optimiser (FlatAdam keeps torch.optim.Adam's
        layout), the schedule, and the explicit step's random stream (seed + step counter on the device: without it a resumed
        run would replay the jitter / noise / normals draws of steps 0..k)."""
        out = {"optimizer": self.optimizer.state_dict(), "scheduler": self.scheduler.state_dict()}
        if self.explicit is not None:
            out["draw_state"] = self.explicit.draw_state.detach().cpu().clone()
        return out

    def load_state_dict(self, sd):
        self.optimizer.load_state_dict(sd["optimizer"])
        self.scheduler.load_state_dict(sd["scheduler"])
        if self.explicit is not None and sd.get("draw_state") is not None:
            # in place: a captured step holds this tensor's address
            self.explicit.draw_state.copy_(sd["draw_state"].to(self.explicit.draw_state.device))

    def step(self, rays, rgbs, alphas, body_model_params, body_model_params_template, fg_points=None, bg_points=None,
             perturb=1.0, frame_idx=None):
        """`body_model_params` is the dict of the batch, or — with a BodyModelParams table and `frame_idx` — replaced
        by the learnable rows of those frames (train.py:330-331)."""
        with self._own_stream():
            loss, details = self._step_body(rays, rgbs, alphas, body_model_params, body_model_params_template, fg_points,
                                            bg_points, perturb, frame_idx)
        # torch's fused Adam updates in place WITHOUT bumping the tensors' version counters; every cached weight pack
        # (training, backward, inference) is keyed by them.  A loop that steps a fused optimiser itself must do the same.
        from .autograd import bump_generation
        bump_generation(self.params)
        return loss, details

    def _step_body(self, rays, rgbs, alphas, body_model_params, body_model_params_template, fg_points, bg_points, perturb,
                   frame_idx, apply=True, at_split=None):
        if self.explicit is not None and self.explicit.supported(rays, body_model_params, frame_idx, fg_points, bg_points):
            # forward, losses and backward as a fixed sequence of the library's launches (fused_step.py): no autograd graph,
            # no framework kernel between the first and the last launch of the step
            loss, details = self.explicit.run(rays, rgbs, alphas, body_model_params, body_model_params_template, fg_points,
                                              bg_points, perturb, frame_idx, at_split=at_split)
            if apply:
                self._apply_gradients()
            return loss, details
        self.begin_step()                                     # grads are views into the (zeroed) flat buffers
        if self.body_model_params is not None and frame_idx is not None:
            body_model_params = self.body_model_params(frame_idx)
        self._attach_prior_points(fg_points, bg_points)
        results = system_forward(self.renderer, self.model, rays, body_model_params, body_model_params_template,
                                 perturb=perturb, chunk=self.hp.chunk)
        loss, details = compute_loss(self.model, self.hp, rgbs, alphas, results, fg_points, bg_points)
        loss.backward()                                       # full buckets are all-reduced while this is still running
        with torch.no_grad():
            key = "rgbs_fine" if "rgbs_fine" in results else "rgbs"
            # train.py:339-344 calls torchmetrics' peak_signal_noise_ratio WITHOUT data_range: the range is the targets' own
            data_range = rgbs.max() - rgbs.min()
            details["psnr"] = 10.0 * torch.log10(data_range ** 2 / F.mse_loss(results[key], rgbs))
        if apply:
            self._apply_gradients()
        return loss.detach(), details

    def _apply_gradients(self):
        self.reducer.finish()
        self.optimizer.step()
