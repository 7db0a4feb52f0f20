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
    ps[0].grad = torch.full((5, 3), float(rank + 1)); ps[1].grad = torch.arange(7.) * (rank + 1)
    if rank == 1:
        ps[2].grad = torch.tensor([4.0, 8.0])            # rank 0 produced no gradient for ps[2] (its rays missed the body)
    n = ana.allreduce_gradients(ps)
    assert n == 24, n                                    # ONE collective over the flat buffer of EVERY parameter
    assert torch.equal(ps[0].grad, torch.full((5, 3), 1.5)) and torch.equal(ps[1].grad, torch.arange(7.) * 1.5)
    assert torch.equal(ps[2].grad, torch.tensor([2.0, 4.0]))      # zeros from rank 0, averaged

    # bucketed + overlapped with backward: gradients accumulate straight into the send buffers; buckets are issued in
    # list order whatever order autograd finishes them in, and a rank without any gradient still joins every collective
    a, b, c = (torch.nn.Parameter(torch.ones(4)) for _ in range(3))
    red = ana.GradientReducer([[a], [b, c]])
    assert red.active
    for it in range(2):
        red.prepare()
        if rank == 0:
            loss = (a * 2.0).sum() + (a * 1.0).sum() + (b * 5.0).sum()      # a used twice: its hook fires once; c unused
            loss.backward()
        # rank 1: no backward at all
        assert red.finish() == 12
        assert torch.equal(a.grad, torch.full((4,), 1.5)) and torch.equal(b.grad, torch.full((4,), 2.5)), (a.grad, b.grad)
        assert torch.equal(c.grad, torch.zeros(4))
    # order: bucket 1 complete before bucket 0 -> still issued 0, 1
    red.prepare()
    (b.sum() + c.sum()).backward()
    assert red._next == 0                                # bucket 1 is ready, bucket 0 is not: nothing issued yet
    (a.sum() * (rank + 1)).backward()
    assert red._next == 2
    red.finish()
    assert torch.equal(a.grad, torch.full((4,), 1.5)) and torch.equal(c.grad, torch.ones(4))
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


def test_bench_self_launch_at_two_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py starts its own ranks as a child
    `torch.distributed.run`, relays rank 0's ONE JSON line and its exit code.  The plumbing-only step (a sleep) runs the
    rendezvous, the verified all-reduce (`collective_ranks`), the barrier-bracketed max-over-ranks timing and the
    per-rank spread without a GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ANR_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--plumbing-only"], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                      # ONE line on stdout, whatever the children printed
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["collective_ranks"] == 2 and line["collective_backend"] == "gloo"
    assert line["steps"] == 3 and line["warmup"] == 1 and line["all_ranks_ok"] is True
    spread = line["rank_ms_per_step"]
    # rank 1 sleeps twice as long as rank 0; the job's clock is the slowest rank's
    assert 9.0 < spread["min"] < spread["max"] and 19.0 < spread["max"] <= line["ms_per_step"] + 1e-6
    assert abs(line["value"] - 2 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]


def test_bench_self_launch_reports_a_failing_rank():
    """a child that cannot start its backend (nccl without a GPU here) must surface as a non-zero exit code and no line"""
    import torch
    if torch.cuda.device_count() > 0:
        return                                            # on a GPU box RCCL comes up: covered by the -m gpu tests
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "ANR_BENCH_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--plumbing-only"], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode != 0 and p.stdout.strip() == ""
