"""Plot -> voxels on the GPU, in memory ("next" row of SURVEY.md 8f: the reference's ``Voxelise``,
``pointstowood/src/preprocessing.py``, which writes ``voxel_*.pt`` files to disk between preprocessing and inference).

Same arithmetic, without the per-voxel ``nonzero`` scans and without the disk round trip: one stable sort of the cell
ids per grid size, voxels are contiguous segments of the sorted order.

* ``ground_normalise``   - height above the per-5 m-XY-cell minimum z (``gpu_ground`` :37-53) -> ``n_z`` column.
* ``quantile_normalize_reflectance`` - rank -> normal quantile -> [-1, 1] (:18-30).
* ``voxelise``           - multi-resolution grid (:55-64), ``min_pts`` filter, cap at ``max_pts`` by reflectance-weighted
                           sampling without replacement / uniform sampling WITH replacement (:116-120, as the reference).
  ``mode="compat"`` bins over every column like the reference (PyG ``voxel_grid`` is handed x, y, z, reflectance,
  ..., n_z - so voxels are also split by height above ground); ``mode="xyz"`` bins over x, y, z only.

The grid step (cell ids over all columns, the stable argsort that groups the points of a voxel, the runs with at least
``min_pts`` points) and the reflectance ranking run on hand-written HIP kernels behind the C ABI: ``p2w_cells_nd``,
``p2w_sort_pairs_u64`` (radix sort), ``p2w_key_runs``.  There is no second path: points on the host are refused like
everywhere else in the package.  (The CPU tests of the host-side logic around the voxeliser - samplers, sharding - swap
``backend`` for ``oracle.preprocess.TensorBackend``, the tensor-operation restatement of the two steps that the GPU tests
compare the kernels with.)  Ground normalisation and the quantile transform themselves are elementwise tensor code.
"""
from __future__ import annotations

import torch

from . import _lib


def ground_normalise(pos, resolution: float = 5.0):
    x, y, z = pos[:, 0].contiguous(), pos[:, 1].contiguous(), pos[:, 2].contiguous()
    # bin edges are built on the host in fp32 exactly like the reference's arange, then moved to the device
    xb = torch.arange(float(x.min()), float(x.max()) + resolution, resolution).to(pos.device)
    yb = torch.arange(float(y.min()), float(y.max()) + resolution, resolution).to(pos.device)
    gi = torch.bucketize(x, xb) * len(yb) + torch.bucketize(y, yb)
    # per-cell minimum straight into a dense table of the (few hundred) 5 m cells: the reference's torch.unique only renumbers them
    zmin = torch.full(((len(xb) + 1) * len(yb) + 1,), float("inf"), device=pos.device).scatter_reduce(0, gi, z, reduce="amin")
    return torch.cat((pos, (z - zmin[gi]).view(-1, 1)), dim=1)


def quantile_normalize_reflectance(refl):
    if torch.isnan(refl).any():
        raise ValueError("Input reflectance tensor contains NaN values.")
    order = backend.argsort_f32(refl)                     # stable ascending (ties keep their input order)
    ranks = torch.empty_like(order)
    ranks[order] = torch.arange(order.numel(), dtype=order.dtype, device=order.device)   # = argsort(order): the inverse permutation
    q = torch.clamp((ranks.float() + 1) / (len(ranks) + 1), 1e-7, 1 - 1e-7)
    n = torch.erfinv(2 * q - 1) * torch.sqrt(torch.tensor(2.0, device=refl.device))
    return 2 * (n - n.min()) / (n.max() - n.min()) - 1


CELL_NONFINITE = (1 << 63) - 1   # P2W_CELL_NONFINITE of include/p2w.h


def _drop_nonfinite_run(cell_sorted, starts, counts):
    """The run of CELL_NONFINITE keys (rows with a non-finite value; it sorts last) is not a voxel."""
    if starts.numel() and int(cell_sorted[starts[-1]]) == CELL_NONFINITE:
        return starts[:-1], counts[:-1]
    return starts, counts


class HipBackend:
    """The voxeliser's two sorting steps on libp2w_gfx950.so (the product path; GPU tensors only)."""

    @staticmethod
    def grid_segments(P, size, min_pts):
        """(order, starts, counts): stable argsort of the cell ids of PyG voxel_grid(P, size) over every column, start and length of
        every run with >= min_pts points: p2w_cells_nd -> p2w_sort_pairs_u64 (stable radix argsort) -> p2w_key_runs."""
        _lib.require_cuda(P)
        L, ptr, stream, check = _lib.lib(), _lib.ptr, _lib.stream, _lib.check
        P = P.contiguous()
        n, D = P.shape
        dev = P.device
        if D > 16:
            raise ValueError("the voxeliser bins at most 16 columns")
        u8 = lambda nbytes: torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
        cell, cell_sorted = torch.empty(n, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.int64, device=dev)
        order = torch.empty(n, dtype=torch.int32, device=dev)
        ws = u8(max(L.p2w_sort_pairs_u64_ws_bytes(n), L.p2w_key_runs_ws_bytes(n)))
        check(L.p2w_cells_nd(ptr(P), n, D, D, float(size), ptr(cell), ptr(ws), ws.numel(), stream()), "p2w_cells_nd")
        check(L.p2w_sort_pairs_u64(ptr(cell), ptr(cell_sorted), None, ptr(order), n, ptr(ws), ws.numel(), stream()), "p2w_sort_pairs_u64")
        starts, counts = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
        n_out = torch.zeros(1, dtype=torch.int32, device=dev)
        check(L.p2w_key_runs(ptr(cell_sorted), n, int(min_pts), ptr(starts), ptr(counts), ptr(n_out), ptr(ws), ws.numel(), stream()),
              "p2w_key_runs")
        k = int(n_out)                                     # the one host sync of a grid size
        return (order.long(), *_drop_nonfinite_run(cell_sorted, starts[:k].long(), counts[:k].long()))

    @staticmethod
    def argsort_f32(x):
        """Stable ascending argsort of a float32 vector (no NaN; -0.0 == +0.0 as in torch.sort) on p2w_sort_pairs_u64: the values'
        bit patterns made order-preserving (sign bit flipped for positives, all bits for negatives) are the radix sort's keys."""
        _lib.require_cuda(x)
        L, ptr, stream, check = _lib.lib(), _lib.ptr, _lib.stream, _lib.check
        n = x.numel()
        bits = (x.to(torch.float32).reshape(-1) + 0.0).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        keys = torch.where(bits >= 0x80000000, 0xFFFFFFFF - bits, bits + 0x80000000).contiguous()
        keys_sorted = torch.empty_like(keys)
        order = torch.empty(n, dtype=torch.int32, device=x.device)
        ws = torch.empty(max(int(L.p2w_sort_pairs_u64_ws_bytes(n)), 256), dtype=torch.uint8, device=x.device)
        check(L.p2w_sort_pairs_u64(ptr(keys), ptr(keys_sorted), None, ptr(order), n, ptr(ws), ws.numel(), stream()), "p2w_sort_pairs_u64")
        return order.long()


backend = HipBackend   # (tests of the host-side logic on the CPU put oracle.preprocess.TensorBackend here)


def voxelise(pc, grid_sizes=(2.0, 4.0), min_pts: int = 128, max_pts: int = 16384, mode: str = "compat", generator=None,
             ground: bool = True, cells: list | None = None):
    """pc: [N, >=4] (x, y, z, reflectance, ...).  Returns (voxels, n_z): ``voxels`` is a list of ``[n, cols+1]`` float32
    tensors on ``pc.device`` in the reference's order (grid size major, ascending cell id).

    ``ground=False`` is the reference's branch for input that already carries an ``n_z`` column (any ``*_ours.ply``
    written by this tool or the reference does): ``gpu_ground`` is skipped, the columns are binned as they come and the
    LAST column is returned as n_z (preprocessing.py:81-86,127) - no second height column is appended.

    ``cells`` (a list, optional) receives one ``(box_lo [v, 3], box_hi [v, 3])`` pair per grid size: the xyz cell of every
    voxel of that grid (all of a voxel's points share it: the grid's own arithmetic - column minimum, fp32 subtract / divide /
    truncate - on the voxel's first row, widened by a thousandth of the cell size) - what the spatially sharded
    back-projection selects voxels by."""
    if mode not in ("compat", "xyz"):
        raise ValueError("mode must be 'compat' or 'xyz'")
    pos = ground_normalise(pc.to(torch.float32)) if ground else pc.to(torch.float32).clone()
    refl_on = not bool(torch.all(pos[:, 3] == 0))
    if refl_on:
        pos[:, 3] = quantile_normalize_reflectance(pos[:, 3].reshape(-1))
    weight = (pos[:, 3] - pos[:, 3].min() + 1e-8) if refl_on else None
    voxels = []
    for size in grid_sizes:
        order, starts, counts = backend.grid_segments(pos if mode == "compat" else pos[:, :3], size, min_pts)
        # One gather puts every voxel's rows next to each other; a voxel that needs neither the max_pts sampling nor
        # the NaN-row filter (almost all of them) is then just a VIEW of that tensor: no per-voxel kernels, one sync.
        gathered = pos[order]
        nan_row = torch.isnan(gathered).any(dim=1)
        nan_before = torch.cumsum(nan_row.to(torch.int64), 0) - nan_row.to(torch.int64)      # NaN rows before row i
        if starts.numel() == 0:
            continue
        ends = starts + counts - 1
        nan_cnt = nan_before[ends] + nan_row[ends].to(torch.int64) - nan_before[starts]
        if cells is not None:
            table = pos if mode == "compat" else pos[:, :3]
            finite = torch.isfinite(table).all(dim=1)                       # rows with a non-finite value take no part in the grid's minima
            lo3 = torch.where(finite[:, None], table[:, :3], torch.full_like(table[:, :3], float("inf"))).amin(dim=0)
            first = gathered[starts, :3]          # (a kept run never starts with a non-finite row: those share the key that sorts last)
            c = ((first - lo3) / float(size)).to(torch.int64).to(torch.float32)
            cells.append((lo3 + c * float(size) - 1e-3 * float(size), lo3 + (c + 1.0) * float(size) + 1e-3 * float(size)))
        for s, c, bad in zip(starts.tolist(), counts.tolist(), nan_cnt.tolist()):
            if c <= max_pts and bad == 0:
                voxels.append(gathered[s:s + c])
                continue
            idx = order[s:s + c]
            if c > max_pts:
                if refl_on:
                    idx = idx[torch.multinomial(weight[idx], max_pts, generator=generator)]
                else:
                    idx = idx[torch.randint(0, c, (max_pts,), generator=generator, device=idx.device)]
            v = pos[idx]
            voxels.append(v[~torch.isnan(v).any(dim=1)])
    return voxels, pos[:, -1]
