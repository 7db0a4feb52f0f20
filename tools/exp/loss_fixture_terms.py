#!/usr/bin/env python3
"""Diagnostic (round 5): which loss term carries the 1e-3..2e-3 relative error of the whole weight gradient of
tests/golden/train_loss.npz's batch against fp64 autograd of the oracle?  One term at a time (the rgb MSE is always on)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import anim_nerf_amd as ana
from anim_nerf_amd import synthetic as syn
from helpers import golden, net_params, oracle_table, InjectedDraws
from oracle import animnerf_oracle as orc
from test_oracle_golden import loss_fixture_draws, loss_fixture_model
from test_gpu_training import _fp64, _hip_fine_samples, _templ, _near_relu_kink

dev = torch.device("cuda:0")
tbl = syn.make_smpl_table(0)
g = golden("train_loss")
F_, H, W, Kc, Kf = int(g["frames"]), int(g["H"]), int(g["W"]), int(g["n_samples"]), int(g["n_importance"])
pose = {k: torch.from_numpy(v) for k, v in syn.animated_pose_params(seed=int(g["pose_seed"]), bs=F_).items()}
templ = _templ()
rays = torch.from_numpy(g["rays"])
tgt_rgb, tgt_a = torch.from_numpy(g["target_rgb"]), torch.from_numpy(g["target_alpha"])
fg, bg = torch.from_numpy(g["fg_points"]), torch.from_numpy(g["bg_points"])
otbl = oracle_table(tbl)
tbl64 = _fp64(otbl)
base = dict(lambda_alphas=0.0, lambda_foreground=0.0, lambda_background=0.0, lambda_normals=0.0)
full = {k: float(g[k]) for k in base}
configs = {"rgb only": base, **{"rgb + " + k[7:]: {**base, k: full[k]} for k in base}, "all": full}
for name, lam in configs.items():
    m = loss_fixture_model(tbl, g, device=dev, mlp_mode="f32")
    m.eval()
    hp = ana.TrainHParams(n_samples=Kc, n_importance=Kf, chunk=int(g["chunk"]), epsilon=float(g["epsilon"]), dis_threshold=float(g["dis_threshold"]), **lam)
    vr = ana.VolumeRenderer(n_coarse=Kc, n_fine=Kf)
    res = ana.system_forward(vr, m, rays.to(dev), {k: v.to(dev) for k, v in pose.items()}, _templ(dev), perturb=0.0, chunk=hp.chunk)
    draws = loss_fixture_draws(g, (1, syn.NUM_VERTS, 3))
    with InjectedDraws(replay=draws):
        loss, details = ana.compute_loss(m, hp, tgt_rgb.to(dev), tgt_a.to(dev), res, fg.to(dev) if lam["lambda_foreground"] else None,
                                         bg.to(dev) if lam["lambda_background"] else None)
    loss.backward()
    Pc = {k: v.double().requires_grad_(True) for k, v in net_params(m.nerf).items()}
    Pf = {k: v.double().requires_grad_(True) for k, v in net_params(m.nerf_fine).items()}
    z_fine = _hip_fine_samples(m, vr, rays, pose, dev)
    out = orc.render_frame(tbl64, Pc, Pf, rays.view(F_, H * W, 8).double(), _fp64(pose), _fp64(templ), n_coarse=Kc, n_fine=Kf,
                           use_unpose=True, chunk=hp.chunk, knn_chunk=512, z_fine=z_fine)
    st = orc.frame_state(tbl64, _fp64(pose), _fp64(templ))
    ref, _ = orc.training_loss(Pc, Pf, out, tgt_rgb.view(F_, H * W, 3).double(), tgt_a.view(F_, H * W, 1).double(), n_samples=Kc,
                               fg_points=fg.double() if lam["lambda_foreground"] else None, bg_points=bg.double() if lam["lambda_background"] else None,
                               verts_template=st["verts_template"], draws=tuple(d.double() for d in draws) if lam["lambda_normals"] else None,
                               epsilon=hp.epsilon, dis_threshold=hp.dis_threshold, **lam)
    ref.backward()
    line = [f"{name:24s} loss {loss.item():.6f} / {ref.item():.6f}"]
    for tag, net, P in (("coarse", m.nerf, Pc), ("fine", m.nerf_fine, Pf)):
        num = den = 0.0
        worst = ("", 0.0)
        for k, p in net.named_parameters():
            if p.grad is None:
                continue
            e = (p.grad.cpu().double() - P[k].grad).pow(2).sum().item()
            d = P[k].grad.pow(2).sum().item()
            num += e; den += d
            if d > 0 and (e / d) ** 0.5 > worst[1]:
                worst = (k, (e / d) ** 0.5)
        line.append(f"{tag} {(num / max(den, 1e-300)) ** 0.5:.2e} (|g| {den ** 0.5:.3e}; worst {worst[0]} {worst[1]:.1e})")
    print("  ".join(line), flush=True)
