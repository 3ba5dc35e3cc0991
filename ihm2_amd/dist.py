"""Multi-GPU sharding of a batch of independent MPC instances.

The path shards trivially: instances do not interact, so GPU g owns the contiguous block
``[g*B/G, (g+1)*B/G)`` of the global batch, track tables are replicated, and there is no data-path
collective.  The only communication is ONE gather of the result block at the end, over
``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on the GPU node, ``gloo`` in the CPU tests).
``torch`` is plumbing here (process group, device buffers); it is imported lazily and never touches the
solver arithmetic.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block split; the first ``total % world`` ranks get one extra instance."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} out of range for world size {world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(array: np.ndarray, world: int, rank: int) -> np.ndarray:
    lo, hi = shard_bounds(array.shape[0], world, rank)
    return array[lo:hi]


def all_gather_blocks(local, total: int, group=None):
    """Gather per-rank result blocks (first axis = local instances) into the global array on every rank.

    ``local`` is a ``torch.Tensor`` living where the process group communicates (GPU for nccl, CPU for
    gloo).  Uneven shards are padded to the largest block for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
    nmax = max(sizes)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local block does not match this rank's shard")
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)
