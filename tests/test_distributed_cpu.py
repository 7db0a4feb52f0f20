"""World-size-2 gloo tests (CPU) of the multi-process plumbing bench.py uses at N > 1:
ray sharding, shard assembly, max-over-ranks timing.  The data path itself has no collective."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import anim_nerf_amd as ana
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"],
                            rank=int(os.environ["RANK"]), world_size=2)
    rank = dist.get_rank()
    n = 1001                                              # odd: uneven shards
    lo, hi = ana.shard_range(n, rank, 2)
    full = torch.arange(n, dtype=torch.float32).view(1, n, 1).repeat(1, 1, 3)
    local = full[:, lo:hi] * 2.0                          # "render" this rank's rays
    out = ana.gather_ray_shards(local, n)
    assert out.shape == (1, n, 3) and torch.equal(out, full * 2.0), "shards must tile the frame exactly"
    t = ana.max_over_ranks(1.0 + rank)                    # rank 1 is slower
    assert t == 2.0, t
    # whole-job throughput as bench.py computes it: units of all ranks / slowest rank's time
    assert abs((2 * n) / t - n) < 1e-9
    dist.barrier()
    dist.destroy_process_group()
    print("ok", rank)
""") % ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_sharding_and_timing():
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), PORT=port, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0 and f"ok {rank}" in out, out


GRAD_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import anim_nerf_amd as ana
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"],
                            rank=int(os.environ["RANK"]), world_size=2)
    rank = dist.get_rank()
    ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
    ps[0].grad = torch.full((5, 3), float(rank + 1)); ps[1].grad = torch.arange(7.) * (rank + 1)   # ps[2]: no grad
    n = ana.allreduce_gradients(ps)
    assert n == 22, n                                                # ONE collective over the flat 22-float buffer
    assert torch.equal(ps[0].grad, torch.full((5, 3), 1.5)) and torch.equal(ps[1].grad, torch.arange(7.) * 1.5)
    assert ps[2].grad is None
    dist.destroy_process_group()
    print("ok", rank)
""") % ROOT


def test_two_rank_gradient_allreduce():
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), PORT=port, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", GRAD_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0 and f"ok {rank}" in out, out
