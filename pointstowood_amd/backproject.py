"""Back-projection of the per-voxel classification onto the original plot points, on the GPU.

Reference: ``PointCloudClassifier`` (``pointstowood/src/predicter.py:107-142``): a KD-tree (pykdtree, CPU) over ALL
classified points (every voxel of every grid size, so most plot points occur several times), each original point takes
its k = 64 (``any_wood == 1``) or 32 nearest classified points, ``pwood`` = median of their probabilities and the label
by the weighted vote / any-wood rule (numba).  Here: the classified points are sorted once into a uniform cell grid
(``p2w_voxel_sample``'s order / keys / grid), the queries into Morton order of the same grid, and the grid-indexed exact
kNN (``p2w_knn_grid`` with ``P2W_SEARCH_BOX``) runs in fp32 on coordinates made local to the cloud.  The reference's tree
measures in float64 on the un-shifted coordinates (``predicter.py:205`` makes ``classified_pc`` float64, pykdtree keeps the
data's type), so ``p2w_knn_refine_f64`` then re-ranks every query's neighbourhood in float64 on the coordinates as given:
the fp32 result bounds the k-th distance, every candidate inside that bound is measured exactly, the k nearest by
(distance, index) are kept - the neighbour SETS are the KD-tree's wherever the k-th and (k + 1)-th distances differ.
``p2w_vote`` does the rest.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import SEARCH_BOX, SEARCH_X_INDEX_IN_W, check, lib, ptr


def _records(xyz: torch.Tensor) -> torch.Tensor:
    out = torch.zeros((xyz.shape[0], 4), dtype=torch.float32, device=xyz.device)
    out[:, :3] = xyz
    return out


def auto_cell(cls_xyz: torch.Tensor, k: int) -> float:
    """Cell size of the search grid from the cloud's own density: 0.25 x the radius scale (k x bounding-box volume / points)^(1/3).
    The search is exact for any cell; its time has a broad minimum there (10 M-point synthetic plot, k = 64: 0.1 m cells 106 ms per
    2 M queries, 0.2 - 0.3 m 56 ms, 0.6 m 70 ms - too small and the region grows through many passes, too large and every query
    scans its neighbours' candidates), and the grid's cell table stays at about 64 / k entries per classified point."""
    ext = (cls_xyz.max(dim=0).values - cls_xyz.min(dim=0).values).clamp(min=1e-3)
    vol = float(ext[0]) * float(ext[1]) * float(ext[2])
    c = 0.25 * (k * vol / max(cls_xyz.shape[0], 1)) ** (1.0 / 3.0)
    return min(max(c, 0.02), 2.0)


def neighbours(cls_xyz: torch.Tensor, query_xyz: torch.Tensor, k: int, cell: float | None = None, chunk: int = 1 << 22,
               table_cells: int = 1 << 30):
    """Exact (float64) k nearest classified points of every query: yields (rows, nbr [len(rows), k] int32, deg) per query
    chunk, neighbours ascending by (distance, index).  ``rows`` are the original query indices of the chunk (queries are
    visited in Morton order).  Coordinates: float32 or float64, taken as they are (a float32 value IS its float64 value)."""
    _lib.require_cuda(cls_xyz, query_xyz)
    L, dev = lib(), cls_xyz.device
    nc, nq = cls_xyz.shape[0], query_xyz.shape[0]
    i32 = dict(dtype=torch.int32, device=dev)
    cls64 = cls_xyz.to(torch.float64)
    origin = cls64.min(dim=0).values                       # the fp32 search runs on coordinates local to the cloud
    cls32 = (cls64 - origin).to(torch.float32)
    if cell is None:
        cell = auto_cell(cls32, k)
    cand = _records(cls32)
    del cls32
    ptr_c = torch.tensor([0, nc], **i32)
    order = torch.empty(nc, **i32)
    skeys = torch.empty(nc, dtype=torch.int64, device=dev)
    grid = torch.zeros(8, dtype=torch.int64, device=dev)
    ws = torch.empty(int(L.p2w_voxel_sample_ws_bytes(max(nc, 1))), dtype=torch.uint8, device=dev)
    idx, ptr_out, batch_out = torch.empty(nc, **i32), torch.empty(2, **i32), torch.empty(nc, **i32)
    check(L.p2w_voxel_sample(ptr(cand), ptr(ptr_c), 1, nc, float(cell), ptr(idx), ptr(ptr_out), ptr(batch_out), ptr(order),
                             ptr(skeys), None, ptr(grid), None, None, ptr(ws), ws.numel(), _lib.stream()), "voxel_sample")
    del idx, batch_out, ws
    rec_c = torch.empty((nc, 4), dtype=torch.float32, device=dev)
    check(L.p2w_index_records(ptr(cand), ptr(order), ptr(ptr_c), 1, nc, ptr(rec_c), _lib.stream()), "index_records")
    del cand
    order64 = order.long()
    cs64 = cls64[order64].contiguous()                     # float64 candidates in the grid's cell-sorted order
    del cls64
    pos_of = torch.empty(nc, **i32)
    pos_of[order64] = torch.arange(nc, **i32)
    del order64
    q64 = query_xyz.to(torch.float64)
    qrec = _records((q64 - origin).to(torch.float32))
    qorder = torch.empty(nq, **i32)
    ws = torch.empty(int(L.p2w_morton_order_ws_bytes(max(nq, 1))), dtype=torch.uint8, device=dev)
    check(L.p2w_morton_order(ptr(qrec), nq, ptr(grid), ptr(qorder), ptr(ws), ws.numel(), _lib.stream()), "morton_order")
    del ws
    qsorted = qrec[qorder.long()].contiguous()
    del qrec
    # cell -> first-candidate table of the plot's grid (one load per search run instead of a bisection of the 10^7 keys); the
    # grid's size is read back once - a plot whose grid would not fit `table_cells` entries is searched by bisection
    dims = grid.cpu()[4:7].tolist()
    ox, oy, oz = origin.cpu().tolist()
    n_cells = int(dims[0]) * int(dims[1]) * int(dims[2])
    cell_start = None
    if 0 < n_cells <= int(table_cells):
        cell_start = torch.empty(n_cells + 1, **i32)
        ws = torch.empty(int(L.p2w_cell_starts_ws_bytes(n_cells)) + 256, dtype=torch.uint8, device=dev)
        check(L.p2w_cell_starts(ptr(skeys), nc, n_cells, ptr(cell_start), ptr(ws), ws.numel(), _lib.stream()), "cell_starts")
        del ws
    for s in range(0, nq, chunk):
        m = min(chunk, nq - s)
        q = qsorted[s:s + m]
        rows = qorder[s:s + m].long()
        ptr_q = torch.tensor([0, m], **i32)
        nbr = torch.empty((m, k), **i32)
        deg = torch.empty(m, **i32)
        check(L.p2w_knn_grid_indexed(ptr(rec_c), ptr(skeys), ptr(ptr_c), ptr(grid), ptr(cell_start), ptr(q), None, ptr(ptr_q), 1, m, k,
                                     ptr(nbr), ptr(deg), None, SEARCH_X_INDEX_IN_W | SEARCH_BOX, _lib.stream()), "knn_grid")
        qs64 = q64[rows].contiguous()
        check(L.p2w_knn_refine_f64(ptr(cs64), ptr(order), ptr(pos_of), ptr(skeys), ptr(cell_start), ptr(grid), ox, oy, oz, ptr(qs64),
                                   m, nc, k, ptr(nbr), ptr(deg), _lib.stream()), "knn_refine_f64")
        yield rows, nbr, deg


def collect_predictions(cls_xyz, cls_pred, cls_prob, query_xyz, any_wood: float = 1.0, cell: float | None = None,
                        chunk: int = 1 << 22):
    """(label [nq], pwood [nq]) float32 - ``PointCloudClassifier.collect_predictions`` (predicter.py:129-142).

    cls_xyz [nc,3], cls_pred [nc] (0/1), cls_prob [nc]: the classified points; query_xyz [nq,3]: the original points
    (float32 or float64; neighbours are the float64 KD-tree's, see ``neighbours``).
    k = 64 when ``any_wood == 1`` else 32 (predicter.py:137)."""
    L = lib()
    k = 32 if any_wood != 1 else 64
    nq, dev = query_xyz.shape[0], query_xyz.device
    pred = cls_pred.to(torch.float32).contiguous()
    prob = cls_prob.to(torch.float32).contiguous()
    label = torch.zeros(nq, dtype=torch.float32, device=dev)
    pwood = torch.zeros(nq, dtype=torch.float32, device=dev)
    if cls_xyz.shape[0] == 0 or nq == 0:
        return label, pwood
    for rows, nbr, deg in neighbours(cls_xyz, query_xyz, k, cell, chunk):
        m = rows.shape[0]
        lab = torch.empty(m, dtype=torch.float32, device=dev)
        pw = torch.empty(m, dtype=torch.float32, device=dev)
        check(L.p2w_vote(ptr(nbr), ptr(deg), k, ptr(pred), ptr(prob), m, float(any_wood), ptr(lab), ptr(pw), _lib.stream()),
              "vote")
        label[rows] = lab
        pwood[rows] = pw
    return label, pwood


def collect_predictions_checked(cls_xyz, cls_pred, cls_prob, query_xyz, any_wood: float = 1.0, cell: float | None = None,
                                chunk: int = 1 << 22):
    """``collect_predictions`` plus, per query, the float64 distance to its k-th neighbour among ``cls_xyz`` (+inf where fewer
    than k candidates exist): what a caller that searched a SUBSET of the classified points needs to prove the result exact
    (pipeline._backproject_spatial: every candidate not in the subset lies farther than that distance)."""
    L = lib()
    k = 32 if any_wood != 1 else 64
    nq, dev = query_xyz.shape[0], query_xyz.device
    pred = cls_pred.to(torch.float32).contiguous()
    prob = cls_prob.to(torch.float32).contiguous()
    label = torch.zeros(nq, dtype=torch.float32, device=dev)
    pwood = torch.zeros(nq, dtype=torch.float32, device=dev)
    dk = torch.full((nq,), float("inf"), dtype=torch.float64, device=dev)
    if cls_xyz.shape[0] == 0 or nq == 0:
        return label, pwood, dk
    c64, q64 = cls_xyz.to(torch.float64), query_xyz.to(torch.float64)
    for rows, nbr, deg in neighbours(cls_xyz, query_xyz, k, cell, chunk):
        m = rows.shape[0]
        lab = torch.empty(m, dtype=torch.float32, device=dev)
        pw = torch.empty(m, dtype=torch.float32, device=dev)
        check(L.p2w_vote(ptr(nbr), ptr(deg), k, ptr(pred), ptr(prob), m, float(any_wood), ptr(lab), ptr(pw), _lib.stream()),
              "vote")
        label[rows] = lab
        pwood[rows] = pw
        full = deg >= k
        last = nbr[:, k - 1].long().clamp(min=0)
        d = (c64[last] - q64[rows]).pow(2).sum(dim=1).sqrt()
        dk[rows] = torch.where(full, d, torch.full_like(d, float("inf")))
    return label, pwood, dk
