"""CPU restatement of the reference voxeliser (``pointstowood/src/preprocessing.py``).  TEST INFRASTRUCTURE.

Follows ``Voxelise.gpu_ground`` (:37-53), ``quantile_normalize_reflectance`` (:18-30), ``grid`` (:55-64) and the
filtering / capping of ``write_voxels`` (:79-127) statement by statement, on CPU tensors, returning the voxels instead
of writing ``voxel_*.pt`` files.  Third-party ``voxel_grid`` = ``oracle.ops.voxel_grid`` (which, like PyG's, bins over
EVERY column it is given - the reference passes x, y, z, reflectance, ..., n_z, so voxels are also split by
height-above-ground and the single maximum-reflectance point gets its own cell).  Random capping uses the given
generator (the reference uses the global RNG), so only voxels within the cap are comparable bit for bit.
PINNED: tests/test_host_golden.py::test_voxeliser_matches_the_reference compares it (and the product's voxeliser) with
voxel files written by the reference's own ``Voxelise.write_voxels`` (tests/golden/host/voxelise.npz; its hard-coded
'cuda' device mapped to the CPU by the generating script).  Tied reflectance values are ranked by an unstable sort there.
"""
import torch

from . import ops

CELL_NONFINITE = (1 << 63) - 1   # P2W_CELL_NONFINITE of include/p2w.h


def cells_nd(P, size):
    """PyG voxel_grid(P, size) with batch=None: every column of P is binned with the same cell size (the arithmetic of
    ``ops.voxel_grid``).  Rows with a non-finite value stay out of the column minima / maxima and get the key CELL_NONFINITE
    (like the product's p2w_cells_nd: the reference's own grid on such input is the integer cast of a NaN, i.e. undefined)."""
    n = P.shape[0]
    Pb = torch.cat([P, torch.zeros((n, 1), dtype=P.dtype, device=P.device)], dim=1)
    S = torch.cat([torch.full((P.shape[1],), float(size), dtype=P.dtype, device=P.device),
                   torch.ones(1, dtype=P.dtype, device=P.device)])
    fin = torch.isfinite(Pb)
    row_ok = fin.all(dim=1)
    inf = torch.full_like(Pb, float("inf"))
    ok2 = row_ok[:, None].expand_as(Pb)
    lo, hi = torch.where(ok2, Pb, inf).min(dim=0).values, torch.where(ok2, Pb, -inf).max(dim=0).values
    cnt = ((hi - lo) / S).to(torch.long) + 1
    stride = torch.ones_like(cnt)
    stride[1:] = torch.cumprod(cnt, 0)[:-1]
    Pb = torch.where(ok2, Pb, lo[None].expand_as(Pb))
    cell = (((Pb - lo[None]) / S[None]).to(torch.long) * stride[None]).sum(dim=1)
    return torch.where(row_ok, cell, torch.full_like(cell, CELL_NONFINITE))


class TensorBackend:
    """Tensor-operation restatement of the two sorting steps of the product's voxeliser (``pointstowood_amd.preprocessing.HipBackend``):
    what the GPU tests compare the HIP kernels with, and the stand-in the CPU tests of the host-side logic (samplers, sharding over
    gloo) put into ``preprocessing.backend`` - the product itself has no tensor path."""

    @staticmethod
    def grid_segments(P, size, min_pts):
        """(order, starts, counts): stable argsort of the cell ids, start and length of every run with >= min_pts points."""
        cell = cells_nd(P, size)
        order = torch.argsort(cell, stable=True)          # points of a voxel keep their original relative order
        cell_sorted = cell[order]
        _, counts = torch.unique_consecutive(cell_sorted, return_counts=True)
        starts = torch.cumsum(counts, 0) - counts
        keep = (counts >= min_pts).nonzero(as_tuple=True)[0]
        starts, counts = starts[keep], counts[keep]
        if starts.numel() and int(cell_sorted[starts[-1]]) == CELL_NONFINITE:   # rows with a non-finite value belong to no voxel
            starts, counts = starts[:-1], counts[:-1]
        return order, starts, counts

    @staticmethod
    def argsort_f32(x):
        return torch.argsort(x, stable=True)


def ground(pos):
    x, y, z = pos[:, 0].contiguous(), pos[:, 1].contiguous(), pos[:, 2].contiguous()
    res = 5.0
    x_bins = torch.arange(float(x.min()), float(x.max()) + res, res)
    y_bins = torch.arange(float(y.min()), float(y.max()) + res, res)
    gi = torch.bucketize(x, x_bins) * len(y_bins) + torch.bucketize(y, y_bins)
    _, inv = torch.unique(gi, return_inverse=True)
    zmin = torch.full((int(inv.max()) + 1,), float("inf")).scatter_reduce(0, inv, z, reduce="amin")
    return torch.cat((pos, (z - zmin[inv]).view(-1, 1)), dim=1)


def quantile_normalize_reflectance(refl):
    _, indices = torch.sort(refl, stable=True)
    ranks = torch.argsort(indices, stable=True)
    q = torch.clamp((ranks.float() + 1) / (len(ranks) + 1), 1e-7, 1 - 1e-7)
    n = torch.erfinv(2 * q - 1) * torch.sqrt(torch.tensor(2.0))
    return 2 * (n - n.min()) / (n.max() - n.min()) - 1


def voxelise(pc, grid_sizes=(2.0, 4.0), min_pts=128, max_pts=16384, generator=None, has_nz=False):
    """pc: [N, >=4] float32 (x, y, z, reflectance, ...).  Returns (list of voxel tensors, n_z).  ``has_nz``: the input frame
    has an 'n_z' column, so gpu_ground is skipped and the LAST column is returned as n_z (preprocessing.py:81-86,127)."""
    pos = pc.float().clone() if has_nz else ground(pc.float())
    refl_on = not bool(torch.all(pos[:, 3] == 0))
    if refl_on:
        pos[:, 3] = quantile_normalize_reflectance(pos[:, 3].view(-1))
    groups = []
    for size in grid_sizes:
        cell = ops.voxel_grid(pos, size)
        for vx in torch.unique(cell):
            idx = (cell == vx).nonzero(as_tuple=True)[0]
            if idx.numel() >= min_pts:
                groups.append(idx)
    weight = (pos[:, 3] - pos[:, 3].min() + 1e-8) if refl_on else None
    out = []
    for idx in groups:
        if idx.numel() > max_pts:
            if refl_on:
                idx = idx[torch.multinomial(weight[idx], max_pts, generator=generator)]
            else:
                idx = idx[torch.randint(0, idx.numel(), (max_pts,), generator=generator)]
        v = pos[idx]
        out.append(v[~torch.isnan(v).any(dim=1)])
    return out, pos[:, -1]
