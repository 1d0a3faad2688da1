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


def all_gather_records(local: torch.Tensor, n_total: int | None = None, pending: list | None = None) -> torch.Tensor:
    """Gather per-image records ``[n_local, K]`` from every rank into ``[n_total, K]`` (rank
    order == frame order for ``shard_range`` shards).  Uneven shards are padded to the largest
    shard for the single ``all_gather_into_tensor`` and trimmed afterwards.

    Even shards (``n_total`` divisible by the world size: the bench's weak-scaling case) cost ONE collective and nothing else --
    no size exchange, no host-to-device copy of a length (a ``torch.tensor([...], device=...)`` per call is a blocking copy: it
    stopped the host from running ahead of the device, 60 us per step on one rank).  ``pending``: when a list is given and the
    shards are even, the collective is issued with ``async_op=True`` and its work handle appended; the returned tensor is
    complete once that handle's ``wait()`` has been called (or the device synchronised) -- the gather of step k then overlaps
    the kernels of step k + 1 (RCCL runs on its own stream).  With uneven shards the call is synchronous whatever ``pending``
    is.  Which of the two paths runs is decided from ``n_total`` and the world size alone, so every rank issues the same
    collective; a rank whose shard does not match an even split raises."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    k = local.shape[1:]
    if n_total is not None and n_total % world == 0:
        # the branch is taken from rank-independent information only: every rank must issue the same collective
        if local.shape[0] != n_total // world:
            raise ValueError(f"all_gather_records: n_total = {n_total} divides evenly over {world} ranks, so every rank must hold "
                             f"{n_total // world} records (shard_range's split); this rank holds {local.shape[0]}")
        out = torch.empty((n_total, *k), dtype=local.dtype, device=local.device)
        if pending is not None:
            pending.append(dist.all_gather_into_tensor(out, local.contiguous(), async_op=True))
        else:
            dist.all_gather_into_tensor(out, local.contiguous())
        return out
    # uneven shards: the size exchange below needs the host anyway, so this path is synchronous -- `pending` receives
    # nothing and the returned tensor is complete (callers that drain `pending` before reading stay correct)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
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


# ---- sharded drivers: every analyzer's per-image record through the same single all-gather --------------------------
def _rank_world() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def sharded_records(n_total: int, local_fn) -> torch.Tensor:
    """Split units ``0 .. n_total-1`` into contiguous blocks over the ranks, let ``local_fn(start, stop)`` produce this
    rank's record tensor ``[stop - start, ...]`` (float64, on the device it computed on), and all-gather the blocks into
    ``[n_total, ...]`` on every rank (rank order == unit order).  Units are EPID frames / WL frames / PF frames -- or
    whole CatPhan volumes (``records_per_unit`` rows each, see ``sharded_volume_records``): SURVEY.md section 8e."""
    rank, world = _rank_world()
    a, b = shard_range(n_total, rank, world)
    local = local_fn(a, b)
    if local.shape[0] != b - a:
        raise ValueError("local_fn must return one record per unit of its shard")
    return all_gather_records(local.contiguous(), n_total)


def sharded_volume_records(n_volumes: int, slices_per_volume: int, local_fn) -> torch.Tensor:
    """CatPhan: split BY VOLUME so that the +-3-slice ``combine_surrounding_slices`` and the per-volume axis fits stay
    local (pylinac/ct.py:3351-3386, 2398-2446).  ``local_fn(v0, v1)`` returns ``[(v1 - v0) * slices_per_volume, K]``
    per-slice records of volumes v0 .. v1-1 -> ``[n_volumes * slices_per_volume, K]`` on every rank."""
    rank, world = _rank_world()
    a, b = shard_range(n_volumes, rank, world)
    local = local_fn(a, b)
    if local.shape[0] != (b - a) * slices_per_volume:
        raise ValueError("local_fn must return slices_per_volume records per volume of its shard")
    k = local.shape[1:]
    out = all_gather_records(local.reshape(b - a, slices_per_volume, *k).contiguous(), n_volumes)
    return out.reshape(n_volumes * slices_per_volume, *k)
