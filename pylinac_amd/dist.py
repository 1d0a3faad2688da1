"""Multi-GPU sharding of independent frames.

One process per GPU (``torch.distributed``, backend ``nccl`` == RCCL over xGMI on ROCm).  Frames
are independent -- the reference literally loops ``for img in self.images: img.analyze()``
(pylinac/winston_lutz.py:1567-1578) -- so the batch index is split into contiguous blocks, there
is NO data-path collective, and the only exchange is ONE all-gather of the per-image scalar
records (a few KB: latency-bound, not link-bound).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block split of ``range(n_total)``; the first ``n_total % world`` ranks get one
    extra frame."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_records(local: torch.Tensor, n_total: int | None = None) -> torch.Tensor:
    """Gather per-image records ``[n_local, K]`` from every rank into ``[n_total, K]`` (rank
    order == frame order for ``shard_range`` shards).  Uneven shards are padded to the largest
    shard for the single ``all_gather_into_tensor`` and trimmed afterwards."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    k = local.shape[1:]
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    if n_total is not None and n_total % world == 0:
        counts = [n_total // world] * world
    else:
        sizes = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(sizes, n_local)
        counts = [int(s.item()) for s in sizes]
    m = max(counts)
    padded = local
    if local.shape[0] != m:
        padded = torch.zeros((m, *k), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    out = torch.empty((world * m, *k), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous())
    if all(c == m for c in counts):
        return out
    return torch.cat([out[r * m : r * m + c] for r, c in enumerate(counts)], dim=0)
