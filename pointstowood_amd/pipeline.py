"""Plot-level inference on one GPU, in memory: voxelise -> classify every voxel -> back-project onto the input points.

The reference does this through the disk (``preprocess`` writes ``voxel_*.pt``, ``SemanticSegmentation`` reads them
back: ``pointstowood/predict.py:116-156``, ``src/predicter.py:148-236``); here every stage stays in HBM:
``preprocessing.voxelise`` -> ``PointBudgetSampler`` + ``collate_device`` -> ``Net.stream`` -> ``backproject``.
"""
from __future__ import annotations

import collections
import time

import torch

from .backproject import collect_predictions
from .dist import batch_cost, gather_rows, partition_batches, slice_for_rank
from .predicter import PointBudgetSampler, collate_device
from .preprocessing import voxelise


def _sync(dev):
    if dev.type == "cuda":   # (the sharding logic is also exercised on the CPU with stand-in stages: tests/test_host_cpu.py)
        torch.cuda.synchronize(dev)


BYTES_PER_POINT = 24 * 1024   # a forward's tensors (47 GiB at 2 M points)


def _free_bytes(dev):
    """Free device memory (None on the CPU: the sharding logic is exercised there with stand-in stages)."""
    return torch.cuda.mem_get_info(dev)[0] if dev.type == "cuda" else None


def default_budget(total_points: int, world: int, free, dist=None, device=None) -> int:
    """Points per forward when the caller names no budget: a fifth (one process) or a tenth (sharded: the LPT plan needs enough
    batches per rank to come out even - tools/scaling_predict.py: 6 batches per rank left 15 - 27 % between the ranks at world 8)
    of a rank's share of the classified points, between 262144
    and 2097152, within 40 % of the free device memory.  The batch list must be IDENTICAL on every rank (each rank takes its
    LPT share of it BY INDEX), so the memory cap comes from the rank with the least free memory (one all-reduce(MIN) of a
    scalar), never from a rank's own reading."""
    budget = min(max(total_points // ((5 if world == 1 else 10) * world), 262144), 2097152)
    if free is not None:
        if world > 1:
            backend = dist.get_backend() if hasattr(dist, "get_backend") else "gloo"
            # (the scalar lives on the device the PLOT is on - a caller that never called torch.cuda.set_device would otherwise
            # put every rank's NCCL tensor on cuda:0 - or on the host for gloo)
            on = (device if device is not None and torch.device(device).type == "cuda" else "cuda") if backend == "nccl" else "cpu"
            t = torch.tensor([int(free)], dtype=torch.int64, device=on)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            free = int(t)
        budget = max(65536, min(budget, int(0.4 * free) // BYTES_PER_POINT))
    return budget


def segment_plot(pc, model, grid_sizes=(2.0, 4.0), min_pts: int = 128, max_pts: int = 16384, is_wood: float = 0.5,
                 any_wood: float = 1.0, max_points: int | None = None, mode: str = "compat", generator=None, stats=None,
                 dist=None, max_voxels: int | None = None, ground: bool = True):
    """pc: [N, >= 4] float tensor on the GPU (x, y, z, reflectance, ...), plot-local coordinates (fp32-safe).
    Returns (n_z [N], label [N], pwood [N]) float32 on the device: the three columns the reference appends
    (``predicter.py:233``).  ``stats`` (dict, optional) receives stage timings and counts.  ``max_points`` /
    ``max_voxels``: budget of one forward; default: a fifth of a rank's share of the classified points, between 262144 and
    2097152 points (10 M-point plot on one GPU: 1.97 s of classification at 131072 points per forward, 1.84 at 524288, 1.78
    at 2097152 / 47 GiB; a rank should still get several batches so that the LPT shares come out even), 1 voxel per 1024 points.  ``ground=False``: ``pc`` already has an n_z column (its last one), see
    ``preprocessing.voxelise``.

    ``dist`` (an initialised ``torch.distributed``, one process per GPU, every rank holding the same ``pc`` and the same
    ``generator`` state): every rank voxelises (cheap, and it makes the voxel list identical everywhere without an
    exchange), classifies its LPT share of the voxel batches, the classified points are all-gathered once, each rank
    back-projects a contiguous slice of the plot and the slices are all-gathered: two data exchanges (plus, with the default
    budget, one all-reduce of a scalar: the ranks' least free memory)."""
    dev = pc.device
    t0 = time.perf_counter()
    vox, n_z = voxelise(pc, tuple(grid_sizes), min_pts, max_pts, mode=mode, generator=generator, ground=ground)
    if stats is not None:
        _sync(dev)
        stats["voxelise_s"], t0 = time.perf_counter() - t0, time.perf_counter()
        stats["voxels"] = len(vox)
    n = pc.shape[0]
    if not vox:   # nothing dense enough to classify (predicter.py would fail on an empty loader)
        return n_z, torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    lengths = [int(v.shape[0]) for v in vox]
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist is not None else (1, 0)
    if max_points is None:
        max_points = default_budget(sum(lengths), world, _free_bytes(dev), dist, dev)
    if max_voxels is None:
        max_voxels = max(1, max_points // 1024)
    batches = list(PointBudgetSampler(lengths, max_points, max_voxels))
    plan, batch_rows = None, [sum(lengths[i] for i in b) for b in batches]
    if world > 1:
        plan = partition_batches([sum(batch_cost(lengths[i]) for i in b) for b in batches], world)   # LPT on est. FLOPs
        batches = [batches[i] for i in plan[rank]]
    if stats is not None:
        stats["max_points"], stats["max_voxels"] = int(max_points), int(max_voxels)
        stats["batch_points"] = [sum(lengths[i] for i in b) for b in batches]     # this rank's forwards
    pending = collections.deque()

    def feed():
        for b in batches:
            d = collate_device([vox[i] for i in b])
            pending.append(d)
            yield d

    xyz, prob = [], []
    for logits in model.stream(feed()):
        d = pending.popleft()
        prob.append(torch.sigmoid(torch.nan_to_num(logits)).reshape(-1))                 # predicter.py:197-199
        # predicter.py:203-211: the un-shifted coordinates are float64 sums of the float32 position and shift
        xyz.append(d.pos.to(torch.float64) + d.local_shift.view(-1, 3)[d.batch].to(torch.float64))
    if xyz:
        cls = torch.cat([torch.cat(xyz), torch.cat(prob)[:, None].to(torch.float64)], 1)
    else:
        cls = torch.zeros((0, 4), dtype=torch.float64, device=dev)
    if world > 1:
        cls = gather_rows(cls, dist)
        # the gather returns rank-major rows; put them back into batch order, the order a single process classifies in: the
        # vote's k nearest classified points break distance ties by index, so only then is the sharded result bit-identical
        start, off = {}, 0
        for r in range(world):
            for b in plan[r]:
                start[b], off = off, off + batch_rows[b]
        if off != cls.shape[0]:
            raise RuntimeError(f"gathered {cls.shape[0]} classified points, the batch plan holds {off}")
        cls = torch.cat([cls[start[b]: start[b] + batch_rows[b]] for b in range(len(batch_rows))]) if batch_rows else cls
    cls_xyz, cls_prob = cls[:, :3].contiguous(), cls[:, 3].to(torch.float32).contiguous()
    cls_pred = (cls_prob >= is_wood).to(torch.float32)                                       # predicter.py:200
    if stats is not None:
        _sync(dev)
        stats["classify_s"], t0 = time.perf_counter() - t0, time.perf_counter()
        stats["classified_points"] = int(cls_prob.numel())
    q0, q1 = slice_for_rank(n, rank, world)
    label, pwood = collect_predictions(cls_xyz, cls_pred, cls_prob, pc[q0:q1, :3], any_wood=any_wood)   # (float64 KD-tree semantics)
    if world > 1:
        both = gather_rows(torch.stack([label, pwood], 1), dist)
        label, pwood = both[:, 0].contiguous(), both[:, 1].contiguous()
    if stats is not None:
        _sync(dev)
        stats["backproject_s"] = time.perf_counter() - t0
    return n_z, label, pwood
