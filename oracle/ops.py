"""Normative CPU definitions of the eight third-party operators on the hot path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Pure PyTorch, fp32, deterministic.

The reference calls these through torch-geometric / torch-cluster / torch-scatter,
none of which is vendored or pinned (``README.md:42-47``); where upstream's CPU and
CUDA builds differ the CUDA rule is adopted because it is the one that is
deterministic and specifiable.  "parity unpinned" at this level - see package doc.

Bit-exactness rules shared with the HIP kernels (``pointstowood_amd/csrc``):

* squared distance is ``((dx*dx)+(dy*dy))+(dz*dz)`` in fp32, every operation
  individually rounded (no FMA contraction, never the |x|^2+|y|^2-2xy form);
* kNN order is lexicographic ``(d2, candidate index)`` - ties go to the lower index;
* ball query keeps the first ``max_num_neighbors`` candidates in ascending
  candidate index with strict ``d2 < r2``, ``r2 = float32(float64(r)*float64(r))``;
* voxel cells use fp32 subtract, fp32 true division and truncation.
"""
from __future__ import annotations

import torch
from torch import Tensor

_CHUNK_PAIRS = 1 << 24  # distance-matrix elements per chunk (64 MiB fp32)


def _ptr_from_sorted_batch(batch: Tensor, num_batches: int) -> Tensor:
    """CSR offsets of a sorted batch vector (torch-cluster does the same with bucketize)."""
    counts = torch.bincount(batch, minlength=num_batches)
    ptr = torch.zeros(num_batches + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(counts, 0)
    return ptr


def _d2_matrix(y: Tensor, x: Tensor) -> Tensor:
    """[m,n] squared distances, diff form, each fp32 op separately rounded."""
    dx = y[:, None, 0] - x[None, :, 0]
    dy = y[:, None, 1] - x[None, :, 1]
    dz = y[:, None, 2] - x[None, :, 2]
    return ((dx * dx) + (dy * dy)) + (dz * dz)


# --------------------------------------------------------------------------- A1
def voxel_grid(pos: Tensor, size: float, batch: Tensor | None = None) -> Tensor:
    """Cell id per point.  Replaces PyG ``voxel_grid`` -> torch-cluster ``grid_cluster``
    as called at ``pointstowood/src/model.py:104``.

    The batch vector is appended as a 4th coordinate with cell size 1, the grid
    origin/extent are the min/max over the WHOLE tensor (so cells depend on batch
    composition), and the id is ``sum_d trunc((p_d-lo_d)/s_d) * stride_d``.
    """
    pos = pos.reshape(pos.shape[0], -1)
    if batch is None:
        batch = torch.zeros(pos.shape[0], dtype=torch.long)
    P = torch.cat([pos, batch.reshape(-1, 1).to(pos.dtype)], dim=1)
    S = torch.cat([torch.full((pos.shape[1],), float(size), dtype=pos.dtype),
                   torch.ones(1, dtype=pos.dtype)])
    lo = P.min(dim=0).values
    hi = P.max(dim=0).values
    cnt = ((hi - lo) / S).to(torch.long) + 1
    stride = torch.ones_like(cnt)
    stride[1:] = torch.cumprod(cnt, 0)[:-1]
    cell = ((P - lo[None, :]) / S[None, :]).to(torch.long)
    return (cell * stride[None, :]).sum(dim=1)


# --------------------------------------------------------------------------- A2
def consecutive_cluster(src: Tensor):
    """(inv, perm): rank of each cell among the sorted unique cells and ONE
    representative point per cell.  Replaces PyG ``consecutive_cluster``
    (``pointstowood/src/model.py:105``).  The representative is the LARGEST point
    index in the cell (CPU ``scatter_`` last-writer rule); ``perm`` is in ascending
    cell id, i.e. batch-major then z, y, x - not in point order.
    """
    uniq, inv = torch.unique(src, sorted=True, return_inverse=True)
    ar = torch.arange(src.shape[0], dtype=torch.long)
    perm = torch.full((uniq.shape[0],), -1, dtype=torch.long)
    perm = perm.scatter_reduce(0, inv, ar, reduce="amax", include_self=True)
    return inv, perm


# --------------------------------------------------------------------------- A3
def radius(x: Tensor, y: Tensor, r: float, batch_x: Tensor | None = None,
           batch_y: Tensor | None = None, max_num_neighbors: int = 32,
           num_workers: int = 1) -> Tensor:
    """Ball query.  Replaces torch-cluster ``radius`` (``pointstowood/src/model.py:118``).
    Returns ``[2,E]`` int64, row 0 = query (y) index, row 1 = candidate (x) index,
    query-major, candidates ascending, at most ``max_num_neighbors`` per query.
    """
    x = x[:, :3].float()
    y = y[:, :3].float()
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    nb = int(max(int(batch_x.max()) if batch_x.numel() else -1,
                 int(batch_y.max()) if batch_y.numel() else -1)) + 1
    px = _ptr_from_sorted_batch(batch_x, nb)
    py = _ptr_from_sorted_batch(batch_y, nb)
    r2 = torch.tensor(float(r) * float(r), dtype=torch.float64).to(torch.float32)
    rows, cols = [], []
    for b in range(nb):
        x0, x1, y0, y1 = int(px[b]), int(px[b + 1]), int(py[b]), int(py[b + 1])
        n, m = x1 - x0, y1 - y0
        if n == 0 or m == 0:
            continue
        step = max(1, _CHUNK_PAIRS // n)
        for s in range(0, m, step):
            e = min(m, s + step)
            hit = _d2_matrix(y[y0 + s:y0 + e], x[x0:x1]) < r2
            keep = hit & (torch.cumsum(hit.to(torch.int32), dim=1) <= max_num_neighbors)
            q, c = torch.nonzero(keep, as_tuple=True)
            rows.append(q + (y0 + s))
            cols.append(c + x0)
    if not rows:
        return torch.zeros(2, 0, dtype=torch.long)
    return torch.stack([torch.cat(rows), torch.cat(cols)], 0)


# --------------------------------------------------------------------------- A4
def knn(x: Tensor, y: Tensor, k: int, batch_x: Tensor | None = None,
        batch_y: Tensor | None = None, cosine: bool = False,
        num_workers: int = 1) -> Tensor:
    """k nearest candidates per query.  Replaces torch-cluster ``knn``
    (``pointstowood/src/model.py:120`` and inside PyG ``knn_interpolate``).
    Returns ``[2,E]`` int64 (row 0 = query index, row 1 = candidate index),
    query-major, ascending ``(d2, index)``; ``min(k, #candidates)`` per query.
    """
    assert not cosine
    x = x[:, :3].float()
    y = y[:, :3].float()
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    nb = int(max(int(batch_x.max()) if batch_x.numel() else -1,
                 int(batch_y.max()) if batch_y.numel() else -1)) + 1
    px = _ptr_from_sorted_batch(batch_x, nb)
    py = _ptr_from_sorted_batch(batch_y, nb)
    rows, cols = [], []
    for b in range(nb):
        x0, x1, y0, y1 = int(px[b]), int(px[b + 1]), int(py[b]), int(py[b + 1])
        n, m = x1 - x0, y1 - y0
        if n == 0 or m == 0:
            continue
        kk = min(k, n)
        cidx = torch.arange(n, dtype=torch.long)
        step = max(1, _CHUNK_PAIRS // (2 * n))
        for s in range(0, m, step):
            e = min(m, s + step)
            d2 = _d2_matrix(y[y0 + s:y0 + e], x[x0:x1])
            # non-negative fp32 -> its bit pattern is monotone; key = (bits, index)
            key = (d2.contiguous().view(torch.int32).to(torch.long) << 32) | cidx[None, :]
            sel = torch.topk(key, kk, dim=1, largest=False, sorted=True).values
            c = sel & 0xFFFFFFFF
            q = torch.arange(s, e, dtype=torch.long)[:, None].expand(-1, kk)
            rows.append(q.reshape(-1) + y0)
            cols.append(c.reshape(-1) + x0)
    if not rows:
        return torch.zeros(2, 0, dtype=torch.long)
    return torch.stack([torch.cat(rows), torch.cat(cols)], 0)


# --------------------------------------------------------------------------- A6
def scatter_max(src: Tensor, index: Tensor, dim: int = 0, out=None, dim_size: int | None = None):
    """Segment max along dim 0.  Replaces torch-scatter ``scatter_max``
    (``pointstowood/src/pointnet.py:122``).  Empty rows are 0; argmax is returned
    for signature compatibility but unused by the reference (-1 placeholder rows)."""
    assert dim == 0
    size = int(index.max()) + 1 if dim_size is None else dim_size
    idx = index.reshape(-1, *([1] * (src.dim() - 1))).expand_as(src)
    res = torch.zeros((size,) + tuple(src.shape[1:]), dtype=src.dtype)
    res = res.scatter_reduce(0, idx, src, reduce="amax", include_self=False)
    arg = torch.full_like(res, -1, dtype=torch.long)
    return res, arg


def segment_max_rows(msg: Tensor, dst: Tensor, num_rows: int) -> Tensor:
    """``aggr='max'`` of PyG ``MessagePassing.propagate`` (``pointstowood/src/pointnet.py:71,108``):
    out[t] = elementwise max of msg[e] over edges with dst_e == t, rows without edges = 0."""
    res = torch.zeros((num_rows, msg.shape[1]), dtype=msg.dtype)
    return res.scatter_reduce(0, dst[:, None].expand_as(msg), msg, reduce="amax", include_self=False)


# --------------------------------------------------------------------------- A7
def global_max_pool(x: Tensor, batch: Tensor, size: int | None = None) -> Tensor:
    """Per-voxel max over rows.  Replaces PyG ``global_max_pool`` (``pointstowood/src/model.py:136``)."""
    nb = int(batch.max()) + 1 if size is None else size
    return segment_max_rows(x, batch, nb)


# --------------------------------------------------------------------------- A8
def knn_interpolate(x: Tensor, pos_x: Tensor, pos_y: Tensor, batch_x: Tensor | None = None,
                    batch_y: Tensor | None = None, k: int = 3, num_workers: int = 1) -> Tensor:
    """Inverse-squared-distance blend of the k nearest coarse features.  Replaces PyG
    ``knn_interpolate`` (``pointstowood/src/model.py:149``): ``w = 1/clamp(d2, 1e-16)``,
    ``out = (sum_c w*x[c]) / (sum_c w)`` with the sums taken in neighbour order from 0."""
    assign = knn(pos_x, pos_y, k, batch_x=batch_x, batch_y=batch_y)
    y_idx, x_idx = assign[0], assign[1]
    diff = pos_x[x_idx, :3] - pos_y[y_idx, :3]
    d2 = (diff * diff).sum(dim=-1, keepdim=True)
    w = 1.0 / torch.clamp(d2, min=1e-16)
    m = pos_y.shape[0]
    num = torch.zeros((m, x.shape[1]), dtype=x.dtype).index_add_(0, y_idx, x[x_idx] * w)
    den = torch.zeros((m, 1), dtype=x.dtype).index_add_(0, y_idx, w)
    return num / den
