"""Worker of tests/test_gpu_training.py::test_two_process_training_step_averages_gradients: two processes (gloo) on ONE
GPU run Trainer.step on different batches; the gradients left in the flat buffers must be the average of the two
batches' single-process gradients, and the parameters must be identical on both ranks afterwards."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import anim_nerf_amd as ana                                        # noqa: E402
from anim_nerf_amd import synthetic as syn                          # noqa: E402
from helpers import golden, seeded_model                            # noqa: E402


def batch(rank, dev):
    gen = torch.Generator().manual_seed(100 + rank)
    pose = {k: torch.from_numpy(v).to(dev) for k, v in syn.animated_pose_params(seed=3 + rank, bs=2).items()}
    c2w, focal, cen = syn.pinhole_camera(8, 8)
    rays = ana.gen_rays(torch.from_numpy(c2w).to(dev), 8, 8, focal.tolist(), 0.1, 10.0, cen.tolist())[None].repeat(2, 1, 1, 1)
    tgt = torch.rand(2, 8, 8, 3, generator=gen).to(dev)
    alp = (torch.rand(2, 8, 8, 1, generator=gen) > 0.5).float().to(dev)
    fg = (torch.rand(2, 64, 3, generator=gen) * 0.4 - 0.2).to(dev)
    bg = (torch.rand(2, 64, 3, generator=gen) * 2 - 1).to(dev)
    return pose, rays, tgt, alp, fg, bg


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    tbl = syn.make_smpl_table(0)
    g = golden("render_cfg3_warp_gain")
    templ = {k: torch.from_numpy(v).to(dev) for k, v in syn.template_pose_params().items()}
    hp = ana.TrainHParams(n_samples=16, n_importance=8, chunk=64, lambda_normals=0.05, lr=1e-3)
    vr = ana.VolumeRenderer(n_coarse=16, n_fine=8)

    # single-process gradients of every rank's batch (plain autograd accumulation, no reducer, no sinks)
    ref = None
    for r in range(world):
        m = seeded_model(tbl, g["seed"], True, g["gain"], g["shift"], device=dev)
        pose, rays, tgt, alp, fg, bg = batch(r, dev)
        torch.manual_seed(500 + r)
        res = ana.system_forward(vr, m, rays, pose, templ, perturb=0.0, chunk=hp.chunk)
        ana.compute_loss(m, hp, tgt, alp, res, fg, bg)[0].backward()
        gr = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        ref = gr if ref is None else {k: ref[k] + gr[k] for k in ref}
    ref = {k: v / world for k, v in ref.items()}

    m = seeded_model(tbl, g["seed"], True, g["gain"], g["shift"], device=dev)
    # (the autograd step: it draws the normal term's points through torch's generator, as the single-process reference above)
    tr = ana.Trainer(m, vr, hp, explicit_step=False)
    assert tr.reducer.active and m.nerf.grad_sink is not None and m.nerf_fine.grad_sink is not None
    pose, rays, tgt, alp, fg, bg = batch(rank, dev)
    torch.manual_seed(500 + rank)
    tr.step(rays, tgt, alp, pose, templ, fg, bg, perturb=0.0)
    for k, p in m.named_parameters():
        if k in ref:
            scale = ref[k].abs().max().item()
            err = (p.grad - ref[k]).abs().max().item()
            assert err <= 2e-5 * scale + 1e-12, (rank, k, err, scale)
    flat = torch.cat([p.detach().reshape(-1) for p in tr.params]).cpu()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert all(torch.equal(both[0], b) for b in both[1:]), "parameters differ between ranks after the step"
    # The graphed step with more than one rank: forward + backward replayed from a HIP graph, ONE all-reduce of the whole flat
    # gradient buffer and Adam after each replay (ANR_GRAPH_SPLIT=1: two graphs, the fine bucket's all-reduce between them).  Same trajectory as the eager step (overlapped bucket all-reduce) on a copy of the model.
    import copy
    # (normals term on: its weight gradients join the flat buffers at the cut between the two graphs; both copies draw the same
    # numbers — the explicit step's stream is a function of the seed at construction and the step count)
    hp2 = ana.TrainHParams(n_samples=16, n_importance=8, chunk=64, lambda_normals=0.05, lr=1e-3)
    base = seeded_model(tbl, g["seed"], True, g["gain"], g["shift"], device=dev)
    me, mg = copy.deepcopy(base), copy.deepcopy(base)
    torch.manual_seed(900)
    te = ana.Trainer(me, vr, hp2)
    torch.manual_seed(900)
    tg = ana.Trainer(mg, vr, hp2, graph=True)
    for it in range(6):
        le, _ = te.step(rays, tgt, alp, pose, templ, fg, bg, perturb=0.0)
        lg, _ = tg.step_graphed(rays, tgt, alp, pose, templ, fg, bg, perturb=0.0)
        assert abs(float(le) - float(lg)) <= 2e-3 * abs(float(le)), (rank, it, float(le), float(lg))
    assert tg._graph is not None and tg._graph_split and te._graph is None
    # ANR_GRAPH_SPLIT=1: the replayed backward cut in two graphs at the fine network's completed bucket (its all-reduce overlaps
    # the second); default: one graph, one all-reduce of the whole gradient buffer
    assert (tg._graph_second is not None) == bool(os.environ.get("ANR_GRAPH_SPLIT"))
    assert tg.reducer.whole is not None and tg.reducer.whole.numel() >= sum(f.numel() for f in tg.reducer.flat)
    assert te.explicit is not None and tg.explicit is not None     # (both ran the explicit step: fused_step.py)
    for (k, a), (_, b) in zip(me.named_parameters(), mg.named_parameters()):
        if a.requires_grad:
            d = (a - b).abs()
            assert d.max() <= 2.0 * hp2.lr and d.mean() <= 0.02 * hp2.lr, (rank, k, float(d.max()), float(d.mean()))
    flat = torch.cat([p.detach().reshape(-1) for p in tg.params]).cpu()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert all(torch.equal(both[0], b) for b in both[1:]), "parameters differ between ranks after the graphed steps"
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: ok")


if __name__ == "__main__":
    main()
