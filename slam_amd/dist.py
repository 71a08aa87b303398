"""Multi-GPU plumbing: one process per GPU, independent streams/pairs sharded over ranks, and the one
collective of the path -- the final trajectory gather.  The reference has no
counterpart (it is single-process); torch.distributed is used only as the RCCL/gloo front-end.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_units: int, rank: int, world: int) -> range:
    """Block distribution of n_units independent streams/pairs over `world` ranks (remainder to the first ranks)."""
    base, rem = divmod(n_units, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def gather_trajectories(local: np.ndarray, device=None) -> np.ndarray:
    """all_gather of per-rank [units, ...] float64 arrays (poses / statuses) -> [sum(units), ...] on every rank.

    Shards may be ragged: sizes are exchanged first, then the padded blocks (one all_gather each, nothing in
    the inner loop).  backend nccl (= RCCL over xGMI) when a process group with GPUs exists, gloo on CPU.
    """
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local, np.float64)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local.copy()
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m,) + local.shape[1:], dtype=torch.float64, device=dev)
    pad[: local.shape[0]] = torch.from_numpy(local).to(dev)
    out = torch.empty((world * m,) + local.shape[1:], dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, pad)
    out = out.cpu().numpy().reshape((world, m) + local.shape[1:])
    return np.concatenate([out[r, : sizes[r]] for r in range(world)], axis=0)
