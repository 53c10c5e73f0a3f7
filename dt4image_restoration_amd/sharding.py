"""Multi-GPU: slices are independent units (no cross-slice term in env.py:74-100), so a job of S slices
is cut into contiguous shards, one per rank, with NO data-path collective.  The only exchange is the
gather of per-slice final PSNR / stop iteration once per episode (RCCL over xGMI on GPUs via
backend "nccl"; gloo on CPU in the tests)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of the slices rank `rank` owns; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_per_slice(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All ranks get the [total, ...] tensor assembled from their shard_range shards (ragged-safe:
    shards are padded to the largest shard for the collective, then trimmed)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(total, r, world) for r in range(world)]
    mx = max(b - a for a, b in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: b - a] for p, (a, b) in zip(parts, sizes)], dim=0)
