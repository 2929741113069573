"""Multi-GPU sharding of voxel batches (one process per GPU, torch.distributed over RCCL/xGMI).

The reference is single-process (``pointstowood/src/predicter.py:150-154``); voxels are classified
independently, the only coupling inside a batch is the batch-global grid origin of ``voxel_grid``, so
the shard unit is a whole voxel batch and the data path has exactly ONE exchange: the gather of the
per-point logits at the end.  Every rank derives the same partition (and therefore everyone's
element counts) from the global list of batch sizes, so no size exchange is needed.
"""
from __future__ import annotations

import torch


# level grid resolutions (model.py:210-212) and the per-unit MACs of SURVEY.md 8(d)
_RES = (0.04, 0.08, 0.16)
_MAC_N, _MAC_M, _MAC_E = 803424, (1245184, 3440640, 12191232), (10496, 74496, 296448)


def batch_cost(n_points: int, volume: float = 8.0, k: int = 32) -> float:
    """Estimated MACs of the forward for one voxel of ``n_points`` points spread over ``volume`` m^3 (default: a 2 m
    voxel): SURVEY.md 8(d)'s formula ``803424 N + 1245184 M1 + 3440640 M2 + 12191232 M3 + 10496 E1 + 74496 E2 + 296448 E3``
    on level sizes estimated by cell occupancy (a level keeps one point per occupied cell: M_l = c_l (1 - exp(-M_{l-1} /
    c_l)) with c_l = volume / res_l^3; for U(2 m, 16384) this gives 15349 / 9773 / 1940 against the measured 15366 /
    10156 / 2185).  The cost is NOT proportional to the point count: small or sparse voxels keep relatively more level-2/3
    points, where a point costs 3-12 M MACs instead of 0.8 M."""
    import math
    n = float(max(0, n_points))
    if n == 0:
        return 0.0
    m, prev = [], n
    for res in _RES:
        cells = max(1.0, volume / res ** 3)
        prev = min(prev, cells * (1.0 - math.exp(-prev / cells)))
        m.append(prev)
    ball = n * (4.0 / 3.0 * math.pi * (2 * _RES[0]) ** 3) / volume + 1.0       # expected points within r = 2 res (+ itself)
    e = (m[0] * min(k, ball), m[1] * min(k, m[0]), m[2] * min(k, m[1]))
    return _MAC_N * n + sum(a * b for a, b in zip(_MAC_M, m)) + sum(a * b for a, b in zip(_MAC_E, e))


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
