"""The eight third-party operators of the reference forward, HIP-backed, with the call
signatures used at the reference's call sites (so its ``src/model.py`` could run over this
module instead of torch-geometric / torch-cluster / torch-scatter):

    voxel_grid(pos, size, batch)                       model.py:104
    consecutive_cluster(src) -> (inv, perm)            model.py:105
    radius(x, y, r, batch_x, batch_y, max_num_neighbors)   model.py:118
    knn(x, y, k, batch_x, batch_y)                     model.py:120
    scatter_max(src, index, dim=0)                     pointnet.py:122
    global_max_pool(x, batch)                          model.py:136
    knn_interpolate(x, pos_x, pos_y, batch_x, batch_y, k)  model.py:149
    pointnet_conv(...)                                 pointnet.py:86-132 (fused: see engine.py)

Every function requires CUDA (MI355X) tensors and libp2w_gfx950.so; there is no fallback.
Index results are int64 like the reference's; the kernels work in int32 internally.
``Net.forward`` does not go through these wrappers (it keeps padded neighbour tables and
fuses the message/aggregate step); they exist for operator-level drop-in use and tests.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, lib, ptr, stream


def _csr(batch, nb):
    b = batch.to(torch.int64).contiguous()
    return torch.searchsorted(b, torch.arange(nb + 1, device=b.device, dtype=torch.int64)).to(torch.int32)


def _num_batches(*batches):
    m = 0
    for b in batches:
        if b is not None and b.numel():
            m = max(m, int(b.max()) + 1)   # host sync, as in torch-cluster's Python wrappers
    return max(m, 1)


def _xyzr(pos):
    _lib.require_cuda(pos)
    n = pos.shape[0]
    out = torch.zeros((n, 4), dtype=torch.float32, device=pos.device)
    out[:, :3] = pos[:, :3].to(torch.float32)
    return out


def _ws(n, device):
    need = int(lib().p2w_voxel_sample_ws_bytes(max(n, 1)))
    if need == 0:
        raise RuntimeError("p2w_voxel_sample_ws_bytes failed")
    return torch.empty(need, dtype=torch.uint8, device=device)


def voxel_grid(pos, size, batch=None):
    _lib.require_cuda(pos)
    n = pos.shape[0]
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long, device=pos.device)
    nb = _num_batches(batch)
    x = _xyzr(pos)
    cell = torch.empty(n, dtype=torch.int64, device=pos.device)
    ws, csr = _ws(n, pos.device), _csr(batch, nb)
    check(lib().p2w_voxel_grid(ptr(x), ptr(csr), nb, n, float(size), ptr(cell), ptr(ws), ws.numel(), stream()),
          "voxel_grid")
    return cell


def consecutive_cluster(src):
    _lib.require_cuda(src)
    n = src.numel()
    src = src.to(torch.int64).contiguous()
    inv = torch.empty(n, dtype=torch.int32, device=src.device)
    perm = torch.empty(n, dtype=torch.int32, device=src.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=src.device)
    ws = _ws(n, src.device)
    check(lib().p2w_consecutive_cluster(ptr(src), n, ptr(inv), ptr(perm), ptr(cnt), ptr(ws), ws.numel(), stream()),
          "consecutive_cluster")
    return inv.to(torch.int64), perm[: int(cnt)].to(torch.int64)


def _search(kind, x, y, arg, batch_x, batch_y, k):
    _lib.require_cuda(x, y)
    dev = x.device
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long, device=dev)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long, device=dev)
    nb = _num_batches(batch_x, batch_y)
    m = y.shape[0]
    nbr = torch.empty((m, k), dtype=torch.int32, device=dev)
    deg = torch.empty(m, dtype=torch.int32, device=dev)
    xx, yy = _xyzr(x), _xyzr(y)
    px, py = _csr(batch_x, nb), _csr(batch_y, nb)
    if kind == "knn":
        bb = torch.empty((lib().p2w_tile_bbox_count(nb, x.shape[0]), 6), dtype=torch.float32, device=dev)
        check(lib().p2w_tile_bbox(ptr(xx), ptr(px), nb, x.shape[0], ptr(bb), stream()), "tile_bbox")
        check(lib().p2w_knn(ptr(xx), ptr(px), ptr(yy), None, ptr(py), nb, m, k, ptr(nbr), ptr(deg), ptr(bb), 0, stream()),
              "knn")
    else:
        check(lib().p2w_ball_query(ptr(xx), ptr(px), ptr(yy), None, ptr(py), nb, m, float(arg), k, ptr(nbr), ptr(deg),
                                   None, 0, stream()), "radius")
    return nbr, deg


def _edges(nbr, deg):
    """padded [m,k] table -> [2,E] (query, candidate), query-major: boolean-mask compaction."""
    m, k = nbr.shape
    mask = torch.arange(k, device=nbr.device)[None, :] < deg[:, None]
    q = torch.arange(m, device=nbr.device)[:, None].expand(m, k)[mask]
    return torch.stack([q, nbr[mask].to(torch.int64)], 0)


def radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32, num_workers=1):
    if not 1 <= max_num_neighbors <= 64:
        raise RuntimeError("max_num_neighbors must be in 1..64")
    return _edges(*_search("radius", x, y, r, batch_x, batch_y, int(max_num_neighbors)))


def knn(x, y, k, batch_x=None, batch_y=None, cosine=False, num_workers=1):
    if cosine:
        raise RuntimeError("cosine distance is not supported")
    if not 1 <= k <= 64:
        raise RuntimeError("k must be in 1..64")   # torch-cluster's CUDA limit is 100
    return _edges(*_search("knn", x, y, None, batch_x, batch_y, int(k)))


def global_max_pool(x, batch, size=None):
    _lib.require_cuda(x, batch)
    nb = _num_batches(batch) if size is None else int(size)
    x = x.to(torch.float32).contiguous()
    out = torch.empty((nb, x.shape[1]), dtype=torch.float32, device=x.device)
    csr = _csr(batch, nb)
    check(lib().p2w_segment_max(ptr(x), x.shape[1], x.shape[1], ptr(csr), nb, ptr(out), stream()), "global_max_pool")
    return out


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    """Sorted-index segment max (the reference's index is the query-major edge target, pointnet.py:122)."""
    assert dim == 0
    res = global_max_pool(src.reshape(src.shape[0], -1), index, size=dim_size)
    return res.reshape((res.shape[0],) + tuple(src.shape[1:])), None


def knn_interpolate(x, pos_x, pos_y, batch_x=None, batch_y=None, k=3, num_workers=1):
    _lib.require_cuda(x, pos_x, pos_y)
    if k > 2 or x.shape[1] % 4:
        raise RuntimeError("knn_interpolate: k <= 2 and a feature width that is a multiple of 4 are supported")
    nbr, deg = _search("knn", pos_x, pos_y, None, batch_x, batch_y, int(k))
    m, F = pos_y.shape[0], x.shape[1]
    out = torch.empty((m, F), dtype=torch.float32, device=x.device)
    xc = x.to(torch.float32).contiguous()
    rc, rf = _xyzr(pos_x), _xyzr(pos_y)   # keep both alive until the launch is enqueued
    check(lib().p2w_interp_concat(ptr(xc), F, ptr(rc), ptr(rf), ptr(nbr), ptr(deg), int(k), None, 0, m,
                                  ptr(out), F, stream()), "knn_interpolate")
    return out
