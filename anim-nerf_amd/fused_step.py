"""The training step of train.py:324-348 as a FIXED SEQUENCE of this library's launches: forward, losses and the whole
backward pass written out by hand, no autograd graph.

Why: the step's dataflow is static (frame state -> warp -> MLP -> compositor, coarse then fine, priors, normals, losses and the
same way back), every backward piece is already a hand-written kernel (`autograd.*Function.backward`), and what torch autograd
adds on top is ~85 framework launches per step — the sums of gradients that reach a tensor twice (`aten::add`), zero-filled
gradient buffers, slices, clones, embedding look-ups, four random-number kernels — plus the host time to walk the graph.  In a
HIP-graph replay each of those is a node; at the per-rank batch of the reference's 8-GPU run (2 frames) the ~210 nodes were
most of the step.  Here the step is ~90 launches, all of this library:

    draws | gather frame params | SMPL (3) | root frame | rays | ober2cano | KNN index | coarse depths |
    warp (3) | compact (3) | MLP | expand | composite          (coarse; the prior points ride along as extra rows)
    sample + merge | warp with reuse (3) | compact (3) | MLP | expand | composite          (fine)
    tangent quads | MLP (tangent) x 2 | losses | losses' backward |
    per network: head grad | activation grads | encode | weight grads (2)          (normals)
    per pass: composite backward | head grad | activation grads | encode | weight grads (2) | d-encoding | encode backward |
              expand | warp backward          (fine, then the merge's backward, then coarse)
    coarse depths' backward | frame chain backward (4) | scatter table grads | [all-reduce] | Adam

Two parts of the step feed nothing that follows them until the very end, and run on STREAMS OF THEIR OWN — parallel branches of
the captured graph: the normals regulariser — all of it: forward, the gradient of its loss term (a function of its own outputs:
the total's upstream gradient is 1), backward, weight gradients into buffers of their own that join the flat gradient with one
`anr_add_inplace` per network — next to the frame set-up and the render passes' forward, whose searches and small launches leave
most of the GPU idle; and the render passes' weight gradients, behind the activation gradients, while the backward chain goes on
towards the points and the poses.  The forward weight packs both branches read are made on the step's stream before the fork,
the backward ones on the weight gradients' stream meanwhile.  4.27 -> 3.66 ms per step at 16 frames, 2.14 -> 1.57 at 2
(`tools/exp/step_timeline.py`: 1.7 ms of the step with one launch running, 1.5 with two, 0.7 with three; launches that share the
GPU slow each other down — the sum of the kernel times goes from 4.3 to 6.9 ms — which is why a third branch, the coarse pass's
backward chain next to the fine pass's, bought nothing at 16 frames).  `ANR_STEP_BRANCHES=0` puts every launch back on the
step's stream (debugging).

Gradients that autograd would sum are accumulated where they are produced: both warp backward passes add into one
dL/d ober2cano and one dL/d rays buffer, the merge's and the coarse depths' backward kernels take their two / three upstream
gradients as separate operands, the weight gradients of every pass add into the network's flat buffer (`GradSink`).

Equality with the autograd step — same loss, same gradients on the same random numbers — is what
tests/test_gpu_training.py::test_explicit_step_equals_the_autograd_step holds it to; the autograd step itself is held to the
oracle's autograd (fp32 and fp64) and to the reference's own compute_loss by the other tests of that file.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib, ops
from .autograd import PARAM_KEYS, _cached_pack, _cached_pack_pair, weights_generation


_TORCH_DRAWS = (torch.rand, torch.randn, torch.randn_like)


class ExplicitTrainStep:
    """Bound to a Trainer; `supported(...)` says whether a call can take this path (otherwise the autograd step runs).
    The step's random numbers come from the library's own counter-based generator (anr_train_draws), not from torch's: a
    caller that substitutes torch.rand / randn / randn_like to control the draws (the tests' InjectedDraws) gets the autograd
    step, which draws through them."""

    def __init__(self, trainer):
        self.tr = trainer
        p0 = trainer.params[0]
        self.dev = p0.device
        # (seed, step counter, tickets): the random numbers of step k are a pure function of (seed, k) — torch.manual_seed
        # before the Trainer is built fixes the run
        self.draw_state = torch.zeros(ops.DRAW_STATE_WORDS, dtype=torch.int64, device=self.dev)
        self.draw_state[0] = torch.initial_seed() & 0x7FFFFFFFFFFFFFFF
        self.last_draws = None         # the random tensors of the last step (t_rand, noise_c, u_fine, noise_f, n0, n1): for tests
        self.last_quads = None         # ... and its tangent quads (coarse, fine)
        self._streams = None           # (the normals branch's stream, the render passes' weight gradients'): see run()
        self._side = None
        self.parallel = os.environ.get("ANR_STEP_BRANCHES", "1") != "0"
        self._wgrad_stream = None      # ... and the render passes' weight gradients'

    # ------------------------------------------------------------------------------------------------------------------
    def supported(self, rays, body_model_params, frame_idx, fg_points, bg_points) -> bool:
        tr, m, vr, hp = self.tr, self.tr.model, self.tr.renderer, self.tr.hp
        if not (rays.is_cuda and torch.is_grad_enabled() and hp.fused_losses and hp.use_unpose):
            return False
        if (torch.rand, torch.randn, torch.randn_like) != _TORCH_DRAWS:
            return False
        bm = m.body_model
        if bm.lbs_weights.shape[1] != 24 or bm.shapedirs.shape[-1] != 10:        # (anr_frame_setup: the SMPL body of the shipped configs)
            return False
        if not (getattr(m, "use_unpose", False) and m.k_neigh == 4 and not m.use_view and hasattr(m, "nerf_fine") and m.nerf_fine is not m.nerf
                and m.evaluate_valid_only and m.skip_far_samples and m.nerf._hip_supported() and m.nerf_fine._hip_supported()):
            return False
        if not (vr.lindisp and vr.n_fine > 0 and vr.n_fine_depth == 0 and not vr.share_fine and hp.n_importance > 0 and not hp.share_fine
                and vr.n_coarse + vr.n_fine <= 256 and getattr(vr, "reuse_coarse_warp", True)):
            return False
        if rays.dim() != 4 or rays.shape[-1] != 8 or rays.shape[1] * rays.shape[2] > hp.chunk or not rays.is_contiguous():
            return False
        # the networks train (their flat gradient buffers are the weight-gradient kernels' destination), or they are FROZEN
        # and the poses alone train: the `_refine` stage of the shipped configs (configs/people_snapshot/*_refine.yaml:
        # pretrained_model_requires_grad False, train.py:433-437)
        frozen = self.frozen_networks()
        if not frozen and (m.nerf.grad_sink is None or m.nerf_fine.grad_sink is None):
            return False
        if frozen and (any(p.requires_grad for net in (m.nerf, m.nerf_fine) for p in net.parameters())
                       or tr.body_model_params is None or frame_idx is None):
            return False
        # the prior points ride along as rows of the MLP passes: one set per frame of the batch, float32 on the rays' device
        # (any other shape goes through the autograd step, which queries them on their own)
        for pts in (fg_points, bg_points):
            if pts is not None and not (torch.is_tensor(pts) and pts.device == rays.device and pts.dtype == torch.float32
                                        and pts.dim() == 3 and pts.shape[0] == rays.shape[0] and pts.shape[2] == 3):
                return False
        # pose refinement through a BodyModelParams table (optim_body_params), or constant poses given as a dict
        if tr.body_model_params is not None and frame_idx is not None:
            t = tr.body_model_params
            if not all(getattr(t, n).weight.is_cuda and getattr(t, n).weight.is_contiguous() for n in t.param_names):
                return False
            ws = [getattr(t, n).weight for n in t.param_names]
            if frozen:
                return all(w.requires_grad for w in ws)
            return all(w.requires_grad for w in ws) or not any(w.requires_grad for w in ws)
        if body_model_params is None or any(torch.is_tensor(v) and v.requires_grad for v in body_model_params.values()):
            return False
        p = body_model_params
        return all(torch.is_tensor(p.get(k)) and p[k].is_cuda and p[k].dtype == torch.float32 and p[k].dim() == 2 and p[k].shape[-1] == d
                   for k, d in (("betas", 10), ("global_orient", 3), ("body_pose", 69), ("transl", 3)))

    def frozen_networks(self) -> bool:
        m = self.tr.model
        return not any(p.requires_grad for p in m.nerf.parameters()) and not any(p.requires_grad for p in m.nerf_fine.parameters())

    # ------------------------------------------------------------------------------------------------------------------
    def _mlp_pass(self, net, mode_id, pts, fg, bg, pack=None, frozen=False, cstate=None):
        """compacted training forward of one network on pts[n,4] (+ the prior points as riders): -> state for the backward
        (out_c[rows, 4]: the output of the listed rows, pos[n + n_r]: a sample's or rider's row)"""
        params = [dict(net.named_parameters())[k] for k in PARAM_KEYS]
        if pack is None:
            pack = _cached_pack(params, mode_id, False)
        index, pos, pts_c, count = ops.compact_ordered_riders(pts, fg, bg, single=True, state=cstate)
        rows = count[1:2]
        # (frozen networks: nothing reads the saved activations — the weight gradients' operands — only the sign bits)
        out_c, act = ops.mlp_forward_save(pack, mode_id, pts_c, False, count=rows, bits_only=frozen)
        # (the output stays compact: the compositor, its backward and the loss read a sample's row through `pos` — no expanded
        # copy, anr_expand_rows, a launch and 32 B per sample in each pass)
        return dict(net=net, params=params, index=index, pos=pos, pts_c=pts_c, count=count, rows=rows, out_c=out_c, act=act)

    def _mlp_backward(self, st, mode_id, g4, want_pts, keep, pack_b=None, frozen=False, wgrad_stream=None):
        """activation, weight (into the network's flat buffer) and — want_pts — point gradients of one compacted pass; g4 = the
        upstream gradient as the backward kernels' operand (the compositor's backward and the loss kernel wrote its rows).
        The weight gradients feed nothing else in the step: they run on a stream of their own (`_wgrad_stream`) behind the
        backward chain, which goes on with the gradient towards the points; `keep` holds what that stream still reads."""
        params, act, rows = st["params"], st["act"], st["rows"]
        if pack_b is None:
            weights_generation(params[0], backward=True)
            pack_b = _cached_pack(params, mode_id, True)
        dact = ops.mlp_backward(pack_b, mode_id, g4, act, count=rows, enc_only=frozen)
        if frozen:                                                   # the gradient towards the points, and nothing else
            return ops.mlp_dpoints(pack_b, mode_id, dact, st["pts_c"], count=rows)
        main = torch.cuda.current_stream(self.dev)
        side = wgrad_stream if wgrad_stream is not None else self._wgrad_stream
        side.wait_stream(main)                                       # (the fork is HERE: the weight gradients wait for dact only)

        def weight_gradients():
            with torch.cuda.stream(side):
                enc = ops.encode64(st["pts_c"], act.dtype, count=rows)
                ops.mlp_wgrad(mode_id, act, dact, enc, g4, accumulate_into=st["net"].grad_sink.flat, count=rows, background=self.parallel)
            keep.append((g4, dact, enc))

        # (issued AFTER the chain's next launches: the graph executor keeps the successor captured first on the queue of the
        # activation gradients and moves the other one to a queue that may be busy with the normals branch's tail — with the
        # weight gradients first, the chain towards the points waited there: 1.64 against 1.7-1.9 ms per step at 2 frames)
        # (the gradient towards the points stays COMPACT — one row per valid sample, anr_mlp_dpoints: d-encoding and the
        # encoding's derivative in one launch — and the warp's backward looks a sample's row up through `pos`)
        out = ops.mlp_dpoints(pack_b, mode_id, dact, st["pts_c"], count=rows) if want_pts else None
        weight_gradients()
        return out

    def _loss_args(self, t, consts):
        a = _lib.AnrLossArgs()
        for k in ("rgb", "acc", "rgb_fine", "acc_fine", "target_rgb", "target_alpha", "quads", "quads_fine"):
            v = t.get(k)
            setattr(a, k, None if v is None else v.data_ptr())
        for k in ("s", "s_fine", "s_count", "s_count_fine"):
            setattr(a, k, t.get(k))                              # raw addresses: column 3 of the pass's rows, its row count
        for k in ("R", "prior_rows", "nv", "normal_sets", "quad_rows", "n_fg", "n_bg", "s_stride", "s_grad_rows", "quad_grad_rows"):
            setattr(a, k, int(consts.get(k, 0)))
        for k in ("k", "delta", "lambda_alphas", "lambda_foreground", "lambda_background", "lambda_normals"):
            setattr(a, k, float(consts.get(k, 0.0)))
        return a

    # ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def run(self, rays, rgbs, alphas, body_model_params, template_params, fg_points, bg_points, perturb, frame_idx, at_split=None):
        """at_split (more than one rank, the graphed step): called once, on the step's stream with every side stream joined, at the
        point where the FINE network's flat gradient is complete (its render pass's and the normals branch's weight gradients
        are in) and the coarse pass's backward has not begun — the caller ends one graph capture there and begins the next,
        so that a replayed step can all-reduce the fine network's bucket while the second graph (coarse backward, pose chain)
        replays (Trainer._step_graphed).  Same launches, same sums."""
        tr, m, vr, hp = self.tr, self.tr.model, self.tr.renderer, self.tr.hp
        lib = _lib.load()
        dev = rays.device
        bm = m.body_model
        bs, H, W = rays.shape[:3]
        R, Kc, Kf = H * W, vr.n_coarse, vr.n_fine
        K = Kc + Kf
        n_c, n_f = bs * R * Kc, bs * R * K
        mode_id = ops.MLP_MODES[m.nerf.mlp_mode] & 0xff
        frozen = self.frozen_networks()
        sinks = () if frozen else (m.nerf.grad_sink, m.nerf_fine.grad_sink)
        refine = False
        table = tr.body_model_params if (tr.body_model_params is not None and frame_idx is not None) else None

        # ---- gradient buffers: the reducer's flat buffers are the destination of everything (zeroed with the step's other
        # fills, below)
        tr.reducer.prepare(zero=False)
        if table is not None:
            refine = all(getattr(table, n).weight.requires_grad for n in table.param_names)

        want_normals = hp.lambda_normals != 0
        jitter = perturb > 0
        noisy = vr.noise_std > 0.0 and perturb > 0
        if not m._same_template(template_params):
            m._set_template(template_params)

        # ---- normals regulariser (models/nerf.py:177-190, train.py:298-309): both networks on the same quads (forward-mode
        # tangents, autograd.QuadSigmaFunction).  It needs the draws and the weights and NOTHING of the render passes — its loss
        # term's gradient is a function of its own outputs (the total's upstream gradient is 1) — and only the loss values and
        # Adam need it: the whole branch, forward, loss gradient, backward and weight gradients (into buffers of their own), runs
        # on a SECOND STREAM next to the frame set-up and the render passes' forward, whose searches and small launches leave
        # most of the GPU idle (a parallel branch of the step's HIP graph).
        main = torch.cuda.current_stream(dev)
        if self.parallel:
            if self._streams is None:
                self._streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
            self._side, self._wgrad_stream = self._streams
        else:                                                        # (debugging, per-kernel timing: every launch on the step's stream)
            self._side = self._wgrad_stream = main
        keep = []
        tan, tan_grads = [], []
        one = self._one()
        # What the frame set-up does NOT need runs next to it, on the weight gradients' stream (idle until the backward pass): the
        # step's random numbers (one launch + the counter's; the coarse depths are their first consumer) and the four weight
        # packs (the first network pass is theirs) — six launches, ~45 us, off the front of the step's chain (round 5).
        # (Forward packs first: a backward pack marks the generation.)
        # Every accumulator and counter of the step is filled HERE, in one launch (anr_zero_segments): the flat gradient
        # buffers, the two neighbour searches' counters, the pose chain's accumulators — seven fills, five of them between
        # launches of the step's chain, before round 5.
        V = bm.lbs_weights.shape[0]
        nets = draws = front_ready = packs_b = packs_b_ready = warp_ws_c = warp_ws_f = acc_buf = frame_ws = quads4 = None
        cstates = None
        n_riders = bs * ((fg_points.shape[1] if fg_points is not None else 0) + (bg_points.shape[1] if bg_points is not None else 0))

        begin = torch.cuda.Event()                                    # (the side stream forks HERE, whatever is issued first)
        begin.record(main)

        def front():
            nonlocal nets, draws, front_ready, packs_b, packs_b_ready, warp_ws_c, warp_ws_f, acc_buf, frame_ws, quads4, cstates
            self._wgrad_stream.wait_event(begin)
            with torch.cuda.stream(self._wgrad_stream):
                fills = list([tr.reducer.whole] if tr.reducer.whole is not None else tr.reducer.flat)
                warp_ws_c, z0 = ops.warp_workspace(bs, R * Kc, dev)
                warp_ws_f, z1 = ops.warp_workspace(bs, R * K, dev)
                fills += [z0, z1]
                # the two compactions' look-back states belong to the step and are zeroed with everything else: neither shared
                # through a process-global table nor dependent on the previous launch having restored them
                cstates = (ops.compact_state(bs * R * Kc + n_riders, dev), ops.compact_state(bs * R * K + n_riders, dev))
                fills += list(cstates)
                acc_buf = frame_ws = None
                if refine:
                    acc_buf = torch.empty(bs * V * 16 + bs * R * 8, dtype=torch.float32, device=dev)
                    frame_ws, z2 = ops.frame_backward_workspace(bs, V, dev)
                    fills += [acc_buf, z2]
                quads4 = None
                if want_normals:                                         # the regulariser's points as tangent-mode quads: the draws write them
                    n_pair = 2 * m.verts_template.numel() // 3
                    n_pad = -(-n_pair // 16) * 16
                    quads4 = torch.empty(4 * n_pad, 4, dtype=torch.float32, device=dev)
                    fills.append(quads4[4 * n_pair:])
                ops.zero_segments(fills)
                draws = ops.train_draws(self.draw_state, n_t=bs * R * Kc if jitter else 0, t_scale=float(perturb),
                                        n_nc=bs * R * Kc if noisy else 0, n_u=bs * R * Kf if jitter else 0, n_nf=bs * R * K if noisy else 0,
                                        noise_scale=float(vr.noise_std), verts_template=m.verts_template if want_normals else None,
                                        point_scale=hp.dis_threshold * 0.5, neighbour_scale=hp.epsilon, quads=quads4)
                both = [[dict(net.named_parameters())[k] for k in PARAM_KEYS] for net in (m.nerf, m.nerf_fine)]
                packs_f = _cached_pack_pair(both[0], both[1], mode_id, False, frozen=frozen)        # (one launch for the two networks)
                nets = [(net, params, pack) for net, params, pack in zip((m.nerf, m.nerf_fine), both, packs_f)]
                front_ready = torch.cuda.Event()                             # the fills, the draws and the forward packs
                front_ready.record(self._wgrad_stream)
                if not frozen:
                    for params in both:
                        weights_generation(params[0], backward=True)
                packs_b = list(_cached_pack_pair(both[0], both[1], mode_id, True, frozen=frozen))
                packs_b_ready = torch.cuda.Event()
                packs_b_ready.record(self._wgrad_stream)

        # ---- per-frame state (models/anim_nerf.py:108-151) from the parameter tables: table rows, SMPL, root frame, rays,
        # ober2cano in two launches (ops.frame_setup), then the KNN index
        rays_w = rays.view(bs, R, 8)
        fs = index = None

        def frame_state():
            nonlocal fs, index
            if table is not None:
                w = {n: getattr(table, n).weight for n in table.param_names}
                tables, fidx = (w["betas"], w["global_orient"], w["body_pose"], w["transl"]), frame_idx
            else:
                p = body_model_params
                tables, fidx = (p["betas"].expand(bs, -1).contiguous(), p["global_orient"].expand(bs, -1).contiguous(),
                                p["body_pose"].expand(bs, -1).contiguous(), p["transl"].expand(bs, -1).contiguous()), None
            fs = ops.frame_setup(tables, fidx, m._chain_consts(), bm,
                                 (m.verts_transform_template, m.shape_offsets_template, m.pose_offsets_template), rays_w)
            m.shape_offsets, m.pose_offsets, m.joints_transform = fs["shape_offsets"], fs["pose_offsets"], fs["A"]
            m.global_transform, m.verts, m.joints, m.verts_transform = fs["g_root"], fs["verts"], fs["joints"], fs["verts_transform"]
            m._knn_index = None
            m._refine = None
            m.ober2cano_transform = fs["ober2cano"]
            index = m.knn_index()

        # (capture order = the order a replayed graph's nodes reach the GPU: ANR_STEP_FRONT_LATE puts the chain's first three
        # launches ahead of the side stream's five)
        if os.environ.get("ANR_STEP_FRONT_LATE"):
            frame_state()
            front()
        else:
            front()
            frame_state()
        betas, pose, transl, A, g_inv = fs["betas"], fs["pose"], fs["transl"], fs["A"], fs["g_inv"]
        rays_b, o2c = fs["rays_body"], fs["ober2cano"]
        lbs, thr = bm.lbs_weights, m.dis_threshold
        self.last_draws = draws
        if want_normals and frozen:
            # frozen networks: the regulariser has no parameter to reach (its points are template vertices + noise, its
            # function the network) — only its VALUE enters the step's loss (train.py:288-309 computes it regardless), so the
            # branch is its two forward launches (the inference kernel in tangent mode keeps nothing)
            pair = draws["pair"]
            n_pad = -(-pair.shape[0] // 16) * 16
            self._side.wait_stream(main)
            self._side.wait_event(front_ready)
            box = {}

            def normals_forward():
                with torch.cuda.stream(self._side):
                    pts4 = box["pts4"] = quads4
                    for net, params, pack in nets:
                        out_t, act_t = ops.mlp_forward_save(pack, mode_id, pts4, sigma_only=True, tangent=True)
                        tan.append((net, params, act_t, out_t.view(n_pad, 4)))
                    box["quads_ready"] = torch.cuda.Event()
                    box["quads_ready"].record(self._side)
                    keep.append(pts4)

            def normals_first_network():
                pass

            def normals_second_network():
                pass
        elif want_normals:
            pair = draws["pair"]
            n_pad = -(-pair.shape[0] // 16) * 16
            consts_n = {"lambda_normals": hp.lambda_normals, "nv": m.verts_template.shape[1], "normal_sets": m.verts_template.shape[0],
                        "quad_rows": n_pad, "delta": 0.02}
            self._side.wait_stream(main)
            self._side.wait_event(front_ready)
            box = {}

            # The branch is ISSUED in two pieces, each right after the step's stream has issued a long launch (the two neighbour
            # searches): a replayed graph's nodes reach the GPU in capture order at ~12 us apiece, so twenty nodes of this branch
            # issued first held the render passes' first launch back by 0.25 ms.  (Its three forward launches issued at the
            # fork, next to the frame set-up's small ones: 3.66 / 1.58-1.83 ms per step against 3.61 / 1.56.)
            def normals_forward():
                with torch.cuda.stream(self._side):
                    pts4 = box["pts4"] = quads4
                    for net, params, pack in nets:
                        out_t, act_t = ops.mlp_forward_save(pack, mode_id, pts4, sigma_only=True, tangent=True)
                        tan.append((net, params, act_t, out_t.view(n_pad, 4)))
                    box["quads_ready"] = torch.cuda.Event()
                    box["quads_ready"].record(self._side)

            def normals_first_network():
                with torch.cuda.stream(self._side):
                    pts4 = box["pts4"]
                    # (the quads' gradient as the backward kernels' operand, four rows (0, 0, 0, .) per quad: no head-gradient launch)
                    d_quads = box["d_quads"] = [torch.empty(4 * n_pad, 4, dtype=torch.float32, device=dev) for _ in range(2)]
                    args_n = self._loss_args({"quads": tan[0][3], "quads_fine": tan[1][3]}, dict(consts_n, quad_grad_rows=1))
                    g_n = _lib.AnrLossGrads()
                    g_n.quads, g_n.quads_fine = d_quads[0].data_ptr(), d_quads[1].data_ptr()
                    _lib.check(lib.anr_train_loss_backward(C.byref(args_n), ops._ptr(one), C.byref(g_n), ops._stream(one)), "anr_train_loss_backward")
                    box["enc4"] = ops.encode64(pts4, tan[0][2].dtype, tangent=True)      # (the same rows for both networks)
                    self._side.wait_event(packs_b_ready)
                    normals_backward(0)
                    keep.append((pts4, box["enc4"], d_quads))

            def normals_backward(i):
                net, params, act_t, _ = tan[i]
                pts4 = box["pts4"]
                g4 = box["d_quads"][i]
                dact = ops.mlp_backward(packs_b[i], mode_id, g4, act_t, sigma_only=True, tangent=True)
                # (sigma only: the tensors behind sigma.bias get nothing from this branch — neither written nor added below)
                tan_grads.append((net, ops.mlp_wgrad(mode_id, act_t, dact, box["enc4"], g4, sigma_only=True, tangent=True, background=self.parallel,
                                                     no_fill=True)))
                keep.append((g4, dact))

            def normals_second_network():
                with torch.cuda.stream(self._side):
                    normals_backward(1)

        # ---- coarse pass
        main.wait_event(front_ready)                                 # the jitter (and, further down, the forward weight packs)
        steps = vr._table(dev, "steps", Kc)
        zc = ops.sample_coarse(rays_b, steps, draws["t_rand"].view(bs * R, Kc) if jitter else None).view(bs, R, Kc)
        pts_c, nidx_c, nw_c = ops.warp_points(index, o2c, lbs, thr, rays=rays_b, z=zc, skip_far=True, neighbours=True, workspace=warp_ws_c)
        if want_normals:
            normals_forward()
            normals_first_network()
        n_r = bs * ((fg_points.shape[1] if fg_points is not None else 0) + (bg_points.shape[1] if bg_points is not None else 0))
        st_c = self._mlp_pass(m.nerf, mode_id, pts_c.view(-1, 4), fg_points, bg_points, nets[0][2], frozen, cstates[0])
        flat_rays = rays_b.view(bs * R, 8)
        noise_c = draws["noise_c"].view(bs * R, Kc) if noisy else None
        w_c, rgb_c, dep_c, acc_c = ops.composite(st_c["out_c"], zc.view(bs * R, Kc), flat_rays, vr.white_bkgd, noise=noise_c,
                                                 want_weights=True, pos=st_c["pos"])
        # ---- fine pass: importance samples + merge, the coarse samples' warp rows copied by the merge's permutation
        u = draws["u_fine"].view(bs * R, Kf) if jitter else vr._table(dev, "u", Kf)
        zs, perm = ops.sample_fine_merge(zc.view(bs * R, Kc), w_c, u, want_perm=True, perm_u8=True)
        zs = zs.view(bs, R, K)
        pts_f, nidx_f, nw_f = ops.warp_points(index, o2c, lbs, thr, rays=rays_b, z=zs, skip_far=True, neighbours=True,
                                              reuse=(pts_c, None, perm, nidx_c, nw_c), workspace=warp_ws_f)
        if want_normals:
            normals_second_network()
        st_f = self._mlp_pass(m.nerf_fine, mode_id, pts_f.view(-1, 4), fg_points, bg_points, nets[1][2], frozen, cstates[1])
        noise_f = draws["noise_f"].view(bs * R, K) if noisy else None
        _, rgb_f, dep_f, acc_f = ops.composite(st_f["out_c"], zs.view(bs * R, K), flat_rays, vr.white_bkgd, noise=noise_f,
                                               want_weights=False, pos=st_f["pos"])

        if want_normals:
            main.wait_event(box["quads_ready"])                      # the loss values read the quads
        self.last_quads = [x[3] for x in tan]
        # ---- losses (train.py:228-322) and their gradients: two launches
        consts = {"R": bs * R, "k": -2.0 / hp.n_samples, "lambda_alphas": hp.lambda_alphas, "lambda_foreground": hp.lambda_foreground,
                  "lambda_background": hp.lambda_background, "lambda_normals": hp.lambda_normals}
        t = {"rgb": rgb_c, "acc": acc_c, "rgb_fine": rgb_f, "acc_fine": acc_f, "target_rgb": rgbs.reshape(-1, 3), "target_alpha": alphas.reshape(-1)}
        assert t["target_rgb"].is_contiguous() and t["target_alpha"].is_contiguous()
        # the upstream gradient of either pass goes straight to the rows of the network's backward operand (round 5: no row per
        # sample in between, no anr_mlp_head_grad): the compositor's backward writes the samples' rows, the loss kernel the prior
        # points'
        g4_c = torch.empty(st_c["pts_c"].shape[0], 4, dtype=torch.float32, device=dev)
        g4_f = torch.empty(st_f["pts_c"].shape[0], 4, dtype=torch.float32, device=dev)
        if n_r:
            # the prior points' sigmas: the last n_r of the listed rows of each pass (count[0] of them, on the device)
            t["s"], t["s_fine"] = st_c["out_c"].data_ptr() + 12, st_f["out_c"].data_ptr() + 12
            t["s_count"], t["s_count_fine"] = st_c["count"].data_ptr(), st_f["count"].data_ptr()
            consts.update(prior_rows=bs, n_fg=fg_points.shape[1] if fg_points is not None else 0,
                          n_bg=bg_points.shape[1] if bg_points is not None else 0, s_stride=4, s_grad_rows=1)
        if want_normals:
            t["quads"], t["quads_fine"] = tan[0][3], tan[1][3]
            consts.update(nv=m.verts_template.shape[1], normal_sets=m.verts_template.shape[0], quad_rows=tan[0][3].shape[0], delta=0.02)
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        ws = ops._LOSS_WS.get(key)
        if ws is None:
            ws = ops._LOSS_WS[key] = torch.zeros(lib.anr_train_loss_ws_floats(), dtype=torch.float32, device=dev)
        vals = torch.empty(12, dtype=torch.float32, device=dev)
        args = self._loss_args(t, consts)
        _lib.check(lib.anr_train_loss(C.byref(args), ops._ptr(ws), ops._ptr(vals), ops._stream(vals)), "anr_train_loss")
        g = _lib.AnrLossGrads()
        d_rgb_c, d_acc_c, d_rgb_f, d_acc_f = (torch.empty_like(x) for x in (rgb_c, acc_c, rgb_f, acc_f))
        g.rgb, g.acc, g.rgb_fine, g.acc_fine = (x.data_ptr() for x in (d_rgb_c, d_acc_c, d_rgb_f, d_acc_f))
        if n_r:
            g.s, g.s_fine = g4_c.data_ptr(), g4_f.data_ptr()
        _lib.check(lib.anr_train_loss_backward(C.byref(args), ops._ptr(one), C.byref(g), ops._stream(vals)), "anr_train_loss_backward")

        # ---- backward: fine pass, the merge, coarse pass, coarse depths, frame chain (the normals' went with their forward)
        d_o2c = d_rays = None
        if refine:
            d_o2c, d_rays = acc_buf[:bs * V * 16].view(bs, V, 4, 4), acc_buf[bs * V * 16:].view(bs, R, 8)
        res = ops.composite_backward(st_f["out_c"], zs.view(bs * R, K), flat_rays, vr.white_bkgd, d_rgb_f, None, d_acc_f,
                                     noise=noise_f, want_dz=refine, pos=st_f["pos"], g4_out=g4_f, count=st_f["count"])
        dz_f = dfar_f = None
        if refine:
            _, dz_f, dfar_f = res
        main.wait_event(packs_b_ready)
        d_pts_f = self._mlp_backward(st_f, mode_id, g4_f, refine, keep, packs_b[1], frozen)
        dz_c_from_fine = None
        if refine:
            dzw_f = ops.warp_backward_acc(d_pts_f, rays_b, zs, o2c, nidx_f, nw_f, d_o2c, d_rays, pos=st_f["pos"])
            dz_c_from_fine = ops.merge_backward2(dzw_f.view(bs * R, K), dz_f, perm, Kc)
        normals_joined = False
        n_sigma = lib.anr_mlp_wgrad_sigma_floats()

        def add_normals():                                           # both networks' share of the regulariser: one launch
            ops.add_segments([(net.grad_sink.flat[:n_sigma], tg[:n_sigma]) for net, tg in tan_grads])

        if at_split is not None and not frozen:
            main.wait_stream(self._wgrad_stream)                     # the fine render pass's weight gradients
            if want_normals:                                         # (both networks' share of the regulariser: 0 + t + r == 0 + r + t)
                main.wait_stream(self._side)
                add_normals()
                normals_joined = True
            at_split()
            self._wgrad_stream.wait_stream(main)                     # (the side stream starts the second capture behind the first's end)
        res = ops.composite_backward(st_c["out_c"], zc.view(bs * R, Kc), flat_rays, vr.white_bkgd, d_rgb_c, None, d_acc_c,
                                     noise=noise_c, want_dz=refine, pos=st_c["pos"], g4_out=g4_c, count=st_c["count"])
        # (the coarse pass's weight gradients on the normals branch's stream, long idle by now, instead of behind the fine
        # pass's on theirs — the weight gradients' stream is what a 16-frame step ends on: 3.24 -> 3.08 ms, same box;
        # all split-K slices for either pass instead of half: 3.10-3.15, gpurun_out/r05/ab_wgrad_streams.txt)
        coarse_wgrad_stream = self._side if (self.parallel and at_split is None and not os.environ.get("ANR_STEP_COARSE_WGRAD_QUEUED")) else None
        d_pts_c = self._mlp_backward(st_c, mode_id, g4_c, refine, keep, packs_b[0], frozen, wgrad_stream=coarse_wgrad_stream)
        if refine:
            _, dz_c, dfar_c = res
            dzw_c = ops.warp_backward_acc(d_pts_c, rays_b, zc, o2c, nidx_c, nw_c, d_o2c, d_rays, pos=st_c["pos"])
            ops.sample_coarse_backward_acc(d_rays.view(bs * R, 8), steps, draws["t_rand"].view(bs * R, Kc) if jitter else None,
                                           dzw_c.view(bs * R, Kc), dz_c, dz_c_from_fine, dfar_c, dfar_f)
            c = m._chain_consts()
            grads = ops.frame_backward(betas, pose, transl, c["J0"], c["JS"], c["parents"], c["lbs_weights"], c["shapedirs"], c["posedirs"],
                                       c["T_template"], rays_world=rays_w, d_o2c=d_o2c, d_rays=d_rays, chain_values=(A, g_inv), workspace=frame_ws)
            wt = {n: getattr(table, n).weight for n in table.param_names}
            ops.scatter_frame_param_grads(frame_idx, grads, wt["global_orient"].shape[0], wt["betas"].shape[0], wt["betas"].grad,
                                          wt["global_orient"].grad, wt["body_pose"].grad, wt["transl"].grad)
        main.wait_stream(self._wgrad_stream)
        if coarse_wgrad_stream is not None:
            main.wait_stream(coarse_wgrad_stream)
        if want_normals and not normals_joined:                      # flat = render passes' + normals' (0 + r + t == 0 + t + r bit for bit)
            main.wait_stream(self._side)
            add_normals()
        if os.environ.get("ANR_STEP_DEBUG_KEEP"):
            # (tools/exp/race_hunt.py: the step's intermediates by name — their addresses are the graph's, the same at every
            # replay — to find WHICH buffer differs when two replays of one step disagree)
            loc = locals()
            names = ("zc", "pts_c", "nidx_c", "nw_c", "w_c", "rgb_c", "acc_c", "zs", "perm", "pts_f", "nidx_f", "nw_f", "rgb_f", "acc_f",
                     "vals", "g4_c", "g4_f", "d_pts_f", "d_pts_c", "acc_buf", "frame_ws", "grads", "dzw_f", "dzw_c", "dz_c_from_fine")
            self.debug_keep = {k: loc[k] for k in names if torch.is_tensor(loc.get(k))}
            self.debug_keep.update({"out_c": st_c["out_c"], "out_f": st_f["out_c"], "cnt_c": st_c["count"], "cnt_f": st_f["count"],
                                    "act_c": st_c["act"], "act_f": st_f["act"], "fs_o2c": o2c, "fs_rays": rays_b, "index": index})
            self.debug_keep.update({f"quads{i}": x[3] for i, x in enumerate(tan)})
            self.debug_keep.update({f"draw_{k}": v for k, v in draws.items() if torch.is_tensor(v)})
        keep.clear()
        for s in sinks:                                              # (three passes each went straight into the flat buffers)
            s.expected = s.done = 0
        details = {k: vals[i] for i, k in enumerate(ops.LOSS_NAMES)}
        if not n_r:
            for k in ("loss_foreground", "loss_background", "loss_foreground_fine", "loss_background_fine"):
                details.pop(k)
        else:
            if fg_points is None:
                details.pop("loss_foreground"), details.pop("loss_foreground_fine")
            if bg_points is None:
                details.pop("loss_background"), details.pop("loss_background_fine")
        if not want_normals:
            details.pop("loss_normals"), details.pop("loss_normals_fine")
        details["psnr"] = vals[11]
        return vals[10], details

    # ------------------------------------------------------------------------------------------------------------------
    def _one(self):
        one = getattr(self, "_one_t", None)
        if one is None or one.device != self.dev:
            one = self._one_t = torch.ones(1, dtype=torch.float32, device=self.dev)
        return one
