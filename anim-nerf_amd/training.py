"""Training step of the rendering path (train.py:102-348 of the reference, minus Lightning).

What is here: the per-step forward (`system_forward`), the reference's photometric / alpha / foreground / background
losses (train.py:228-286), Adam + the polynomial schedule (utils/__init__.py:33-58), and the data-parallel gradient
reduction that replaces Lightning's `strategy='dp'` (config.py:77): one process per GPU, one all-reduce of the
flattened gradient per step (RCCL over xGMI; 2 x 592,388 fp32 = 4.7 MB).

What is not: the normals regulariser (train.py:288-309, second-order autograd through `NeRF.get_normal`) and
gradients into the SMPL parameters (`optim_body_params`).  Both raise if requested.
"""
from __future__ import annotations

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
    lambda_normals: float = 0.0          # reference default 0.01; needs second-order autograd (not built)
    max_epochs: int = 30
    poly_exp: float = 0.9


def compute_loss(anim_nerf, hp: TrainHParams, rgbs, alphas, results, fg_points=None, bg_points=None):
    """train.py:228-286 (rgb MSE, alpha L1, foreground / background sigma priors), coarse and fine."""
    if hp.lambda_normals != 0:
        raise NotImplementedError("normals regulariser (train.py:288-309) needs second-order autograd; set lambda_normals=0")
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
    if hp.use_unpose and fg_points is not None:
        for tag, use_fine in (("", False),) + ((("_fine", True),) if fine else ()):
            s = anim_nerf.query_canonical_space(fg_points, use_fine=use_fine, only_sigma=True)
            details["loss_foreground" + tag] = torch.mean(torch.exp(k * torch.relu(s)))
            loss = loss + hp.lambda_foreground * details["loss_foreground" + tag]
    if hp.use_unpose and bg_points is not None:
        for tag, use_fine in (("", False),) + ((("_fine", True),) if fine else ()):
            s = anim_nerf.query_canonical_space(bg_points, use_fine=use_fine, only_sigma=True)
            details["loss_background" + tag] = torch.mean(1 - torch.exp(k * torch.relu(s)))
            loss = loss + hp.lambda_background * details["loss_background" + tag]
    return loss, details


def allreduce_gradients(params, world: Optional[int] = None):
    """Average gradients over ranks with ONE collective on a flat buffer (DataParallel's reduce_add, train.py:454-455,
    as an all-reduce).  No-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch._utils._flatten_dense_tensors(grads)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world or dist.get_world_size())
    for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(f)
    return flat.numel()


class Trainer:
    """optimizer + schedule + step; `step(batch)` mirrors AnimNeRFSystem.training_step (train.py:324-348)."""

    def __init__(self, anim_nerf, volume_renderer, hp: TrainHParams):
        self.model, self.renderer, self.hp = anim_nerf, volume_renderer, hp
        for name, p in anim_nerf.named_parameters():          # SMPL member params are unused by the forward
            if name.startswith("body_model."):
                p.requires_grad_(False)
        self.params = [p for p in anim_nerf.parameters() if p.requires_grad]
        self.optimizer = torch.optim.Adam(self.params, lr=hp.lr, eps=1e-8, weight_decay=0)
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(
            self.optimizer, lambda epoch: (1 - epoch / hp.max_epochs) ** hp.poly_exp)

    def step(self, rays, rgbs, alphas, body_model_params, body_model_params_template, fg_points=None, bg_points=None,
             perturb=1.0):
        self.optimizer.zero_grad(set_to_none=True)
        results = system_forward(self.renderer, self.model, rays, body_model_params, body_model_params_template,
                                 perturb=perturb, chunk=self.hp.chunk)
        loss, details = compute_loss(self.model, self.hp, rgbs, alphas, results, fg_points, bg_points)
        loss.backward()
        allreduce_gradients(self.params)
        self.optimizer.step()
        with torch.no_grad():
            key = "rgbs_fine" if "rgbs_fine" in results else "rgbs"
            details["psnr"] = -10.0 * torch.log10(F.mse_loss(results[key], rgbs))
        return loss.detach(), details
