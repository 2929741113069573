"""Multi-GPU sharding of voxel batches (one process per GPU, torch.distributed over RCCL/xGMI).

The reference is single-process (``pointstowood/src/predicter.py:150-154``); voxels are classified
independently, the only coupling inside a batch is the batch-global grid origin of ``voxel_grid``, so
the shard unit is a whole voxel batch and the data path has exactly ONE exchange: the gather of the
per-point logits at the end.  Every rank derives the same partition (and therefore everyone's
element counts) from the global list of batch sizes, so no size exchange is needed.
"""
from __future__ import annotations

import torch


def batch_cost(n_points: int) -> float:
    """Relative cost of a voxel (the forward is dominated by per-point dense layers)."""
    return float(n_points)


def partition_batches(costs, world: int):
    """Longest-processing-time assignment of batches to ranks: returns ``world`` lists of batch ids.
    Deterministic (ties -> lower batch id, lower rank) so every rank computes the same plan."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    plan = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        plan[r].append(i)
        load[r] += costs[i]
    for p in plan:
        p.sort()
    return plan


def gather_logits(logits: torch.Tensor, dist, counts=None, group=None):
    """All-gather of per-point logits.  ``counts[r]`` = elements contributed by rank r (None = equal on
    all ranks).  Ragged contributions are padded to the maximum so ONE collective moves them."""
    world = dist.get_world_size(group)
    if counts is None:
        out = [torch.empty_like(logits) for _ in range(world)]
        dist.all_gather(out, logits.contiguous(), group=group)
        return torch.cat(out)
    m = max(counts)
    buf = torch.zeros(m, dtype=logits.dtype, device=logits.device)
    buf[: logits.numel()] = logits
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)])


def gather_rows(rows: torch.Tensor, dist, group=None):
    """All-gather of ``[n_r, c]`` row blocks whose lengths differ per rank and are NOT known in advance (e.g. the
    classified points of each rank's voxel batches): one small all-gather of the lengths, one of the padded blocks.
    Returns the concatenation in rank order, identical on every rank."""
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    counts = [int(v) for v in ns]
    m = max(counts)
    buf = torch.zeros((m,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    buf[: rows.shape[0]] = rows
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)])


def slice_for_rank(n: int, rank: int, world: int):
    """Contiguous share of ``n`` items for ``rank`` (sizes differ by at most one; every item has one owner)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
