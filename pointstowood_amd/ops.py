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
    MessagePassing.propagate(edge_index, x=, pos=)  (aggr=max)   pointnet.py:108 -> ``MessagePassing`` below
    PointNetConv(local_nn, ...)(x, (pos, pos[idx]), edge_index)   pointnet.py:19-132, called at model.py:123 (fused kernel)

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
    if not 1 <= max_num_neighbors <= 100:   # torch-cluster's limit (65 .. 100 run on the one-thread-per-query path)
        raise RuntimeError("max_num_neighbors must be in 1..100")
    return _edges(*_search("radius", x, y, r, batch_x, batch_y, int(max_num_neighbors)))


def knn(x, y, k, batch_x=None, batch_y=None, cosine=False, num_workers=1):
    if cosine:
        raise RuntimeError("cosine distance is not supported")
    if not 1 <= k <= 100:
        raise RuntimeError("`k` needs to smaller than or equal to 100")   # torch-cluster's TORCH_CHECK (65 .. 100: one thread per query)
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
    """PyG's signature and default (k = 3); any 1 <= k <= 100 and any feature width (rows are padded to a multiple of 4
    floats for the kernel's 16-byte accesses)."""
    _lib.require_cuda(x, pos_x, pos_y)
    if not 1 <= k <= 100:
        raise RuntimeError("knn_interpolate: k must be in 1..100")
    nbr, deg = _search("knn", pos_x, pos_y, None, batch_x, batch_y, int(k))
    m, F0 = pos_y.shape[0], x.shape[1]
    F = (F0 + 3) // 4 * 4
    xc = x.to(torch.float32)
    if F != F0:
        xc = torch.nn.functional.pad(xc, (0, F - F0))
    xc = xc.contiguous()
    out = torch.empty((m, F), dtype=torch.float32, device=x.device)
    rc, rf = _xyzr(pos_x), _xyzr(pos_y)   # keep both alive until the launch is enqueued
    check(lib().p2w_interp_concat(ptr(xc), F, ptr(rc), ptr(rf), ptr(nbr), ptr(deg), int(k), None, 0, m,
                                  ptr(out), F, stream()), "knn_interpolate")
    return out if F == F0 else out[:, :F0].contiguous()


# --------------------------------------------------------------------------- the 8th operator
class MessagePassing(torch.nn.Module):
    """The part of PyG's ``MessagePassing`` the reference uses (``pointnet.py:19,71,108``): ``propagate(edge_index, **kw)``
    with ``flow='source_to_target'`` and ``aggr='max'``.  ``edge_index[0]`` = source j, ``edge_index[1]`` = target i,
    grouped by target (as ``radius`` / ``knn`` return them).  ``message``'s parameters are collected the PyG way:
    ``<name>_j`` = ``kw[name]`` (its first element if a pair) gathered by source, ``<name>_i`` = (second element) by
    target, ``edge_index_i`` / ``edge_index_j`` the index rows; the messages are max-aggregated per target
    (``scatter_max``, HIP), targets without edges get 0.  The reference's own ``PointNetConv`` subclass runs on this
    base unchanged; ``PointNetConv`` below is the fused replacement of the whole layer."""

    def __init__(self, aggr="max", flow="source_to_target", **kw):
        super().__init__()
        if aggr != "max" or flow != "source_to_target":
            raise NotImplementedError("only aggr='max', flow='source_to_target' (what the reference uses)")
        self.aggr, self.flow = aggr, flow

    def reset_parameters(self):
        pass

    def propagate(self, edge_index, size=None, **kw):
        import inspect
        j, i = edge_index[0], edge_index[1]
        args = {}
        for name in inspect.signature(self.message).parameters:
            if name == "edge_index_i":
                args[name] = i
            elif name == "edge_index_j":
                args[name] = j
            elif name.endswith("_j") or name.endswith("_i"):
                v = kw[name[:-2]]
                side = 0 if name.endswith("_j") else 1
                v = v[side] if isinstance(v, (tuple, list)) else v
                args[name] = None if v is None else v[j if side == 0 else i]
            else:
                args[name] = kw[name]
        msg = self.message(**args)
        if size is not None:
            n_dst = size[1] if isinstance(size, (tuple, list)) else size
        else:
            n_dst = None
            for v in kw.values():   # number of targets = rows of the second element of any pair argument
                if isinstance(v, (tuple, list)) and v[1] is not None:
                    n_dst = v[1].shape[0]
                    break
                if torch.is_tensor(v):
                    n_dst = v.shape[0]
            if n_dst is None:
                n_dst = int(i.max()) + 1
        return scatter_max(msg, i, dim=0, dim_size=n_dst)[0]


def _local_nn_weights(local_nn):
    """(W1, b1, W2, b2, bn_scale, bn_shift) of ``MLP([F_in + 4, C1, C2])`` (model.py:198-202): Lin + ReLU, Lin + ReLU + BN."""
    try:
        lin1, lin2, bn = local_nn[0][0], local_nn[1][0], local_nn[1][2]
    except (TypeError, IndexError) as e:
        raise NotImplementedError("the fused PointNetConv supports local_nn = MLP([F_in + 4, C1, C2]) as the reference builds it") from e
    if bn.training:
        raise RuntimeError("pointstowood_amd.ops.PointNetConv is inference-only: call .eval()")
    s = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps))
    t = bn.bias.double() - bn.running_mean.double() * s
    return lin1.weight, lin1.bias, lin2.weight, lin2.bias, s.float(), t.float()


class PointNetConv(MessagePassing):
    """Drop-in for the reference's ``src.pointnet.PointNetConv`` (``pointnet.py:19-132``; built at ``model.py:94``, called
    at ``:123``) in eval mode: same constructor, same parameter names (``local_nn.*``), same
    ``forward(x, (pos_src, pos_dst), edge_index)`` with ``pos`` = [n, 4] (sf-scaled xyz, reflectance) - message MLP +
    max aggregation by ONE fused HIP kernel (``p2w_sa_conv``, fp32 MFMA: the [E, C] edge tensors are never formed) after a
    hoisted layer-1 GEMM (``p2w_gemm``).  Edges must be grouped by target with at most ``P2W_MAX_K_CONV`` = 32 per target
    (``radius(max_num_neighbors=32)`` / ``knn(k=32)`` as the reference calls them); ``global_nn`` is applied afterwards as
    in the reference; ``add_self_loops`` must be False (the reference passes False)."""

    def __init__(self, local_nn=None, global_nn=None, add_self_loops=True, **kw):
        self.radius = kw.pop("radius", None)
        kw.setdefault("aggr", "max")
        super().__init__(**kw)
        self.local_nn, self.global_nn, self.add_self_loops = local_nn, global_nn, add_self_loops

    @torch.no_grad()
    def forward(self, x, pos, edge_index):
        import ctypes as C
        if self.add_self_loops:
            raise NotImplementedError("add_self_loops=True is not supported (the reference builds the layer with False)")
        x_src = x[0] if isinstance(x, (tuple, list)) else x
        pos_src, pos_dst = pos if isinstance(pos, (tuple, list)) else (pos, pos)
        _lib.require_cuda(x_src, pos_src, pos_dst, edge_index)
        dev = pos_src.device
        W1, b1, W2, b2, bn_s, bn_t = _local_nn_weights(self.local_nn)
        n_src, M, F_in = pos_src.shape[0], pos_dst.shape[0], x_src.shape[1]
        C1, C2 = W1.shape[0], W2.shape[0]
        if W1.shape[1] != F_in + 4 or pos_src.shape[1] != 4 or C1 % 4:
            raise RuntimeError("PointNetConv: local_nn must take F_in + 4 inputs, pos must be [n, 4], C1 a multiple of 4")
        j, i = edge_index[0].to(torch.int64), edge_index[1].to(torch.int64)
        E = j.numel()
        deg = torch.bincount(i, minlength=M)
        if E and (int(deg.max()) > 32 or bool((i[1:] < i[:-1]).any())):
            raise RuntimeError("PointNetConv: edges must be grouped by target, at most 32 per target")
        start = torch.cumsum(deg, 0) - deg
        nbr = torch.full((M, 32), -1, dtype=torch.int32, device=dev)
        if E:
            nbr[i, torch.arange(E, device=dev) - start[i]] = j.to(torch.int32)
        deg = deg.to(torch.int32)
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        # hoisted layer 1: P = x_src W1x^T + b1 once per source point
        Np, Kp = _lib.packed_dims(C1, F_in)
        Wx = torch.zeros((Np, Kp), dtype=torch.float32, device=dev)
        Wx[:C1, :F_in] = f32(W1[:, :F_in])
        xs = f32(x_src)
        if F_in % 4:
            xs = torch.nn.functional.pad(xs, (0, 4 - F_in % 4)).contiguous()
        # (+ M zero rows: an empty neighbour slot is addressed through the target's own record, which follows the sources)
        P = torch.zeros((n_src + M, C1), dtype=torch.float32, device=dev)
        b1d = f32(b1)
        ep = _lib.Epilogue(ptr(b1d), None, None, None, None, None, 0, 0, 0, 0, 0)
        check(lib().p2w_gemm(ptr(xs), xs.shape[1], ptr(Wx), n_src, C1, F_in, C.byref(ep), ptr(P), C1, stream()), "PointNetConv hoist")
        _, C1p = _lib.packed_dims(C2, C1)
        w1r4 = torch.zeros((4, C1p), dtype=torch.float32, device=dev)
        w1r4[:, :C1] = f32(W1[:, F_in:F_in + 4]).t()
        N2p, _ = _lib.packed_dims(C2, C1)
        W2p = torch.zeros((N2p, C1p), dtype=torch.float32, device=dev)
        W2p[:C2, :C1] = f32(W2)
        # sources and targets in one record array (the kernel addresses a target through an index into the sources)
        rec = torch.cat([f32(pos_src), f32(pos_dst)], 0)
        idx = torch.arange(n_src, n_src + M, dtype=torch.int32, device=dev)
        one, zb = torch.ones(1, dtype=torch.float32, device=dev), torch.zeros(M, dtype=torch.int32, device=dev)
        b2d, sd, td = f32(b2), f32(bn_s), f32(bn_t)
        out = torch.empty((M, C2), dtype=torch.float32, device=dev)
        check(lib().p2w_sa_conv(ptr(P), C1, ptr(rec), ptr(idx), ptr(zb), ptr(one), ptr(nbr), ptr(deg), 32, M, ptr(w1r4),
                                ptr(W2p), C1, C2, ptr(b2d), ptr(sd), ptr(td), ptr(out), C2, stream()), "PointNetConv")
        if self.global_nn is not None:
            out = self.global_nn(out)
        return out
