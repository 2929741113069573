"""Plot-level inference on one GPU, in memory: voxelise -> classify every voxel -> back-project onto the input points.

The reference does this through the disk (``preprocess`` writes ``voxel_*.pt``, ``SemanticSegmentation`` reads them
back: ``pointstowood/predict.py:116-156``, ``src/predicter.py:148-236``); here every stage stays in HBM:
``preprocessing.voxelise`` -> ``PointBudgetSampler`` + ``collate_device`` -> ``Net.stream`` -> ``backproject``.
"""
from __future__ import annotations

import collections
import time

import torch

from .backproject import collect_predictions, collect_predictions_checked
from .dist import batch_cost, gather_logits, gather_rows, partition_batches, slice_for_rank
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
                 dist=None, max_voxels: int | None = None, ground: bool = True, shard: str = "spatial", halo: float = 1.0):
    """pc: [N, >= 4] float tensor on the GPU (x, y, z, reflectance, ...), plot-local coordinates (fp32-safe).
    Returns (n_z [N], label [N], pwood [N]) float32 on the device: the three columns the reference appends
    (``predicter.py:233``).  ``stats`` (dict, optional) receives stage timings and counts.  ``max_points`` /
    ``max_voxels``: budget of one forward; default: a fifth of a rank's share of the classified points, between 262144 and
    2097152 points (10 M-point plot on one GPU: 1.97 s of classification at 131072 points per forward, 1.84 at 524288, 1.78
    at 2097152 / 47 GiB; a rank should still get several batches so that the LPT shares come out even), 1 voxel per 1024 points.  ``ground=False``: ``pc`` already has an n_z column (its last one), see
    ``preprocessing.voxelise``.

    ``dist`` (an initialised ``torch.distributed``, one process per GPU, every rank holding the same ``pc`` and the same
    ``generator`` state): every rank voxelises (cheap, and it makes the voxel list identical everywhere without an
    exchange) and classifies its LPT share of the voxel batches.  ``shard="spatial"`` (default): the per-point probabilities
    - 4 bytes per classified point, the ONLY thing a rank cannot compute itself - are all-gathered once; the back-projection
    is owned spatially (rank r takes the r-th share of the plot points in ascending x and builds its search grid over the
    classified points of the voxels within ``halo`` metres of its slab only; a query whose k-th neighbour could lie beyond
    the slab's covered range is repeated against a 4 x wider set, at most against everything - no further exchange); the
    per-point results are all-gathered once.  ``shard="slices"``: round 5's flow (all classified points to every rank, plot
    slices in input order).  Both give the single-process result bit for bit; with the default budget one all-reduce of a
    scalar (the ranks' least free memory) precedes them."""
    dev = pc.device
    t0 = time.perf_counter()
    cells = []
    vox, n_z = voxelise(pc, tuple(grid_sizes), min_pts, max_pts, mode=mode, generator=generator, ground=ground, cells=cells)
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
    all_batches = batches
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

    spatial = world > 1 and shard == "spatial"
    xyz, prob = [], []
    for logits in model.stream(feed()):
        d = pending.popleft()
        prob.append(torch.sigmoid(torch.nan_to_num(logits)).reshape(-1))                 # predicter.py:197-199
        if not spatial:
            # predicter.py:203-211: the un-shifted coordinates are float64 sums of the float32 position and shift
            xyz.append(d.pos.to(torch.float64) + d.local_shift.view(-1, 3)[d.batch].to(torch.float64))
    if spatial:
        return _backproject_spatial(pc, vox, cells, lengths, all_batches, plan, batch_rows, prob, n_z, is_wood, any_wood, halo, dist, stats, t0)
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


def _x_slab_owners(x, world: int, bins: int = 4096):
    """Ownership of the plot points by slabs in x with (nearly) equal point counts: slab boundaries at the bin edges of a
    `bins`-bin histogram of x where the running count passes r / world (no sort of the plot), owner = the slab a point's x falls
    into.  Returns (owner [n] uint8-like tensor, the per-rank index lists in input order).  Every rank computes the same thing."""
    x = x.to(torch.float32)
    n = x.numel()
    lo, hi = (float(v) for v in torch.aminmax(x)) if n else (0.0, 0.0)
    if not (hi > lo) or world == 1:
        owner = torch.zeros(n, dtype=torch.int64, device=x.device)
    else:
        width = (hi - lo) / bins
        b = ((x - lo) / width).to(torch.int64).clamp_(0, bins - 1)
        cum = torch.cumsum(torch.bincount(b, minlength=bins), 0)
        targets = torch.tensor([n * r // world for r in range(1, world)], device=x.device)
        edges = torch.searchsorted(cum, targets)            # slab r ends behind bin edges[r]
        owner = torch.searchsorted(edges, b, right=False)   # bins <= edges[0] -> 0, ...
    order = torch.argsort(owner.to(torch.uint8) if world <= 256 else owner, stable=True)      # one pass instead of `world` masks
    counts = torch.bincount(owner, minlength=world).cpu().tolist()
    return owner, list(order.split(counts))


def _classified_xyz(voxels):
    """Un-shifted float64 coordinates of the classified points of `voxels` (a list of voxel tensors), as the classify loop
    forms them (predicter.py:203-211: float32 centred position + float32 shift, added in float64).  Per-voxel arithmetic
    (collate_device: sequential per-segment mean), so the bits do not depend on which voxels are collated together - any rank
    can compute any voxel's classified coordinates from the voxel list it holds anyway."""
    d = collate_device(voxels)
    return d.pos.to(torch.float64) + d.local_shift.view(-1, 3)[d.batch].to(torch.float64)


def _backproject_spatial(pc, vox, cells, lengths, batches, plan, batch_rows, prob_parts, n_z, is_wood, any_wood, halo, dist, stats, t0):
    """The sharded back-projection with spatial ownership (see segment_plot).  Exchanges: ONE all-gather of the float32
    probabilities (sizes known to everyone from the plan), ONE all-gather of (label, pwood) (sizes known from the plot size)."""
    dev, n = pc.device, pc.shape[0]
    world, rank = dist.get_world_size(), dist.get_rank()
    # -- exchange 1: every classified point's probability, in the order ONE process classifies in (batch by batch) ----------
    mine = torch.cat(prob_parts) if prob_parts else torch.zeros(0, dtype=torch.float32, device=dev)
    counts = [sum(batch_rows[b] for b in plan[r]) for r in range(world)]
    if mine.numel() != counts[rank]:
        raise RuntimeError(f"rank {rank} classified {mine.numel()} points, the batch plan holds {counts[rank]}")
    gathered = gather_logits(mine.to(torch.float32), dist, counts)
    start, off = {}, 0
    for r in range(world):
        for b in plan[r]:
            start[b], off = off, off + batch_rows[b]
    prob_all = torch.cat([gathered[start[b]: start[b] + batch_rows[b]] for b in range(len(batch_rows))]) if batch_rows else gathered
    if stats is not None:
        _sync(dev)
        stats["classify_s"], t0 = time.perf_counter() - t0, time.perf_counter()
        stats["classified_points"] = int(prob_all.numel())
        stats["exchange_bytes"] = [4 * max(counts) * world]
    # -- who owns what: plot points in ascending x, rank r the r-th share; voxels by their bounding boxes ---------------------
    tick = time.perf_counter()

    def lap(key):
        nonlocal tick
        if stats is not None:
            _sync(dev)
            now = time.perf_counter()
            stats.setdefault("backproject_parts_s", {})[key] = round(stats.get("backproject_parts_s", {}).get(key, 0.0) + now - tick, 4)
            tick = now
    owner, own_lists = _x_slab_owners(pc[:, 0], world)
    own = own_lists[rank]
    # row of every voxel's first classified point in the single-process order, voxel ids in that order
    vox_order = [v for b in batches for v in b]
    vstart, acc = [0] * len(lengths), 0
    for v in vox_order:
        vstart[v], acc = acc, acc + lengths[v]
    vlen = torch.tensor(lengths, dtype=torch.int64, device=dev)
    # a box that holds every voxel: its xyz grid cell (preprocessing.voxelise hands the cells out with the voxels)
    box_lo = torch.cat([a for a, _ in cells]).to(torch.float32)
    box_hi = torch.cat([b for _, b in cells]).to(torch.float32)
    if box_lo.shape[0] != len(lengths):
        raise RuntimeError("voxeliser cells and voxel list disagree")
    vx_lo, vx_hi = box_lo[:, 0].cpu().tolist(), box_hi[:, 0].cpu().tolist()
    order_t = torch.tensor(vox_order, dtype=torch.int64, device=dev)
    vstart_t = torch.tensor(vstart, dtype=torch.int64, device=dev)
    lap("ownership")
    k = 32 if any_wood != 1 else 64
    label = torch.zeros(own.numel(), dtype=torch.float32, device=dev)
    pwood = torch.zeros(own.numel(), dtype=torch.float32, device=dev)
    SLACK = 1e-3      # a classified coordinate is the voxel's raw one up to float32 rounding of the centring: 1 mm covers it

    def search(sel, queries):
        """Exact (label, pwood, k-th distance) of `queries` among the classified points of the voxels `sel` (ids in the
        single-process order, so that distance ties break as they do there)."""
        cxyz = _classified_xyz([vox[v] for v in sel])
        ls = vlen[torch.tensor(sel, dtype=torch.int64, device=dev)]
        first = torch.cumsum(ls, 0) - ls
        rows = torch.repeat_interleave(vstart_t[torch.tensor(sel, dtype=torch.int64, device=dev)] - first, ls) + torch.arange(int(ls.sum()), device=dev)
        cprob = prob_all[rows].contiguous()
        cpred = (cprob >= is_wood).to(torch.float32)                                   # predicter.py:200
        return collect_predictions_checked(cxyz, cpred, cprob, queries, any_wood=any_wood)
    if own.numel() and prob_all.numel():
        qxyz = pc[own, :3]
        qx = qxyz[:, 0].to(torch.float64)
        xlo, xhi = float(qx.min()), float(qx.max())
        tiers = []
        # tier 1: the slab.  Candidates = every voxel that reaches into [xlo - halo, xhi + halo] (all classified points inside that
        # range are among them); a query is settled when its k-th neighbour is STRICTLY closer than the nearer end of the range
        lo, hi = (xlo - halo if rank > 0 else -float("inf")), (xhi + halo if rank < world - 1 else float("inf"))
        sel = [v for v in vox_order if vx_hi[v] >= lo - SLACK and vx_lo[v] <= hi + SLACK]
        if sel:
            lab, pw, dk = search(sel, qxyz)
            ok = (dk < torch.minimum(qx - lo, hi - qx)) if len(sel) < len(vox_order) else torch.ones_like(dk, dtype=torch.bool)
            label[ok], pwood[ok] = lab[ok], pw[ok]
            todo = (~ok).nonzero(as_tuple=True)[0]
        else:
            todo = torch.arange(own.numel(), device=dev)
        tiers.append((len(sel), int(own.numel()), int(todo.numel())))
        lap("tier1")
        # second tier: the few queries left (isolated points whose k-th classified neighbour is metres away).  Their k-th distance
        # among the slab's candidates BOUNDS the true one from above (a subset's k-th neighbour is never closer), so the voxels that
        # reach into the ball of that radius around the query hold its true k nearest: one more search settles them exactly.  A
        # query that found fewer than k candidates at all has no bound: its radius grows 4 x per round, at most to everything.
        R = dk[todo].clone() if sel else torch.full((todo.numel(),), float("inf"), dtype=torch.float64, device=dev)
        grow = max(4.0 * halo, 1.0)
        while todo.numel():
            Rq = torch.where(torch.isfinite(R), R * (1.0 + 1e-9), torch.full_like(R, grow)).to(torch.float32) + SLACK
            qq = qxyz[todo].to(torch.float32)
            near = torch.zeros(len(lengths), dtype=torch.bool, device=dev)
            for part, rad in zip(qq.split(4096), Rq.split(4096)):      # [m, V] box-reaches-ball tests (by the ball's bounding cube)
                hit = ((box_lo[None, :, :] <= part[:, None, :] + rad[:, None, None]) & (box_hi[None, :, :] >= part[:, None, :] - rad[:, None, None])).all(dim=2)
                near |= hit.any(dim=0)
            sel = order_t[near[order_t]].cpu().tolist()
            everything = len(sel) == len(vox_order)
            if sel:
                lab, pw, dk2 = search(sel, qxyz[todo])
                ok = torch.ones_like(dk2, dtype=torch.bool) if everything else (torch.isfinite(R) | (dk2 < grow))
                label[todo[ok]], pwood[todo[ok]] = lab[ok], pw[ok]
                left, R = todo[~ok], torch.where(torch.isfinite(dk2), dk2, R)[~ok]
            else:
                left = todo
            tiers.append((len(sel), int(todo.numel()), int(left.numel())))
            todo = left
            grow *= 4.0
            if everything:
                break
        lap("further_tiers")
        if stats is not None:
            stats["backproject_tiers"] = tiers      # (voxels in the candidate set, queries asked, queries left) per tier
    # -- exchange 2: the per-point results, rank-major; back to input order ----------------------------------------------------
    sizes = [int(o.numel()) for o in own_lists]
    both = gather_logits(torch.stack([label, pwood], 1).reshape(-1), dist, [2 * c for c in sizes]).view(-1, 2)
    where = torch.cat(own_lists)
    out_label = torch.empty(n, dtype=torch.float32, device=dev)
    out_pwood = torch.empty(n, dtype=torch.float32, device=dev)
    out_label[where], out_pwood[where] = both[:, 0], both[:, 1]
    if stats is not None:
        _sync(dev)
        stats["backproject_s"] = time.perf_counter() - t0
        stats["exchange_bytes"].append(8 * max(sizes) * world)
    return n_z, out_label, out_pwood
