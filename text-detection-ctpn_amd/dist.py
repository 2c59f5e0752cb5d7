"""Multi-GPU plumbing: one process per GPU, images sharded embarrassingly, weights sent once.

The path has no per-batch exchange: every image is an independent unit end to end, so the only
collective is ONE broadcast of the 71.57 MB fp32 weight arena at start-up. bench.py sends it over RCCL / xGMI through the
C ABI (ctpn_comm_unique_id + ctpn_broadcast_weights_rank); torch.distributed (backend "gloo") is the side channel for the
128-byte RCCL id, the barrier and the MAX / gather of a few scalars for reporting -- and the fallback carrier of the arena
(broadcast_arena on CPU tensors) where RCCL cannot be used (two ranks on ONE device in the single-GPU test of the N > 1 path).
"""
import os

import numpy as np


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_items, rank, world):
    """Contiguous slice [lo, hi) of n_items for `rank`; the first n_items % world ranks take one extra."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_process_group(backend):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend)
    return dist


def broadcast_arena(arena_or_none, device, src=0):
    """Rank `src` passes the flat fp32 arena (numpy); every rank gets a torch tensor on `device` holding it.
    With world size 1 (or no process group) this is just the host->device copy."""
    import torch
    import torch.distributed as dist
    from .weights import WEIGHT_FLOATS
    t = torch.empty((WEIGHT_FLOATS,), dtype=torch.float32, device=device)
    is_src = (not dist.is_initialized()) or dist.get_rank() == src
    if is_src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(arena_or_none, dtype=np.float32).reshape(-1)))
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def broadcast_bytes(payload_or_none, nbytes, src=0):
    """Rank `src` passes `nbytes` bytes; every rank returns them (a CPU uint8 broadcast: the RCCL unique id's side channel)."""
    import torch
    import torch.distributed as dist
    t = torch.zeros((nbytes,), dtype=torch.uint8)
    is_src = (not dist.is_initialized()) or dist.get_rank() == src
    if is_src:
        t.copy_(torch.from_numpy(np.frombuffer(bytes(payload_or_none), dtype=np.uint8).copy()))
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return bytes(t.numpy().tobytes())


def min_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def max_over_ranks(value, device):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(values, device):
    """Every rank passes the same number of floats; every rank gets a list (one entry per rank) of lists. Tens of bytes, for
    reporting only (per-rank step time, broadcast time: a straggler shows up in the SCALE record instead of hiding in the MAX)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, t)
        return [[float(x) for x in o.tolist()] for o in outs]
    return [[float(x) for x in t.tolist()]]


def barrier():
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
