"""CPU, world_size 2 over gloo: the N > 1 path of bench.py (shard assignment, one weight broadcast, MAX-over-ranks
timing) is correct by construction; the same functions run over RCCL ("nccl") on the GPUs."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

import ctpn_amd  # noqa: F401
from ctpn_amd import dist as D


def test_shard_ranges_partition_the_work():
    for n in (0, 1, 5, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ctpn_amd
    from ctpn_amd import dist as D
    dist = D.init_process_group("gloo")
    arena = None
    if rank == 0:
        arena = np.arange(ctpn_amd.WEIGHT_FLOATS, dtype=np.float32) * np.float32(0.5)
    t = D.broadcast_arena(arena, "cpu", src=0)
    ok = bool(t[12345].item() == 12345 * 0.5 and t[-1].item() == np.float32((ctpn_amd.WEIGHT_FLOATS - 1) * 0.5))
    mx = D.max_over_ranks(1.0 + rank, "cpu")
    lo, hi = D.shard_range(64, rank, world)
    per_rank = D.gather_over_ranks([10.0 + rank, 0.5 * rank], "cpu")
    uid = D.broadcast_bytes(bytes(range(128)) if rank == 0 else None, 128, src=0)       # the RCCL unique id's side channel
    ok = ok and uid == bytes(range(128)) and D.min_over_ranks(1.0 if rank == 0 else 0.0) == 0.0
    D.barrier()
    q.put((rank, ok, mx, lo, hi, per_rank))
    dist.destroy_process_group()


def test_broadcast_and_max_reduce_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [2.0, 2.0]
    assert [(r[3], r[4]) for r in res] == [(0, 32), (32, 64)]
    assert all(r[5] == [[10.0, 0.0], [11.0, 0.5]] for r in res)          # every rank sees every rank's numbers


def test_host_thread_budget_divides_the_node_between_ranks():
    """ctpn_host_thread_budget (pure function of the C ABI): the host workers of one ctx -- per-image connector work of
    ctpn_detect_collect, staging copies -- are the node's cores divided by the ranks on it, clamped to [1, 32]; the
    reference runs that work on one Python thread. 8 ranks x budget must never oversubscribe the node."""
    from ctpn_amd import _binding as B
    for cores in (8, 64, 96, 128, 256, 384):
        for world in (1, 2, 4, 8):
            b = B.host_thread_budget(cores, world, 0)
            assert 1 <= b <= 32 and b * world <= max(cores, world)
    assert B.host_thread_budget(256, 8, 0) == 32 and B.host_thread_budget(256, 1, 0) == 32 and B.host_thread_budget(64, 8, 0) == 8
    assert B.host_thread_budget(4, 8, 0) == 1                       # more ranks than cores: one worker (the caller itself)
    assert B.host_thread_budget(256, 8, 12) == 12                   # CTPN_HOST_THREADS override
    assert B.host_thread_budget(0, 0, 0) == 1


def _bench(args, env=None, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout, cwd=root)


def test_bench_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus N` with no launcher around it starts N ranks itself (torch.distributed.run, loopback
    rendezvous on a free port); under a launcher the world size must agree with --gpus -- a 1-rank run can no longer be labelled N."""
    p = _bench(["--gpus", "2", "--print-launch"])
    assert p.returncode == 0, p.stderr[-2000:]
    seen = sorted(l for l in p.stdout.splitlines() if l.startswith("bench.py launch:"))
    assert len(seen) == 2 and "rank 0/2 local 0 master 127.0.0.1:" in seen[0] and "rank 1/2 local 1 master 127.0.0.1:" in seen[1]
    assert seen[0].split("master ")[1] == seen[1].split("master ")[1]            # one rendezvous
    p = _bench(["--print-launch"])
    assert p.returncode == 0 and "rank 0/1" in p.stdout
    p = _bench(["--gpus", "2", "--print-launch"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "must agree" in p.stderr
    p = _bench(["--gpus", "1", "--print-launch"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert p.returncode != 0 and "must agree" in p.stderr
