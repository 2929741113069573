"""The voxeliser's grid step on the GPU (reference pointstowood/src/preprocessing.py:55-64): the hand-written
radix sort, the n-column cell ids and the run filter through the C ABI, against tensor-level definitions, and the whole
``preprocessing.voxelise`` on GPU tensors (HIP kernels) against the same call on host tensors (tensor operations; that path is
pinned to the reference's own voxeliser by tests/test_host_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu


def _L():
    from pointstowood_amd._lib import lib
    return lib()


@pytest.mark.parametrize("n,bits,with_vals", [(1, 64, False), (63, 8, True), (4096, 16, False), (4097, 20, True), (100003, 37, False),
                                              (1 << 20, 63, True), (300000, 0, False), (70000, 9, True),
                                              # beyond 1 048 576 keys the histogram scan is two-level (rs_scan_tile / _sums / _add): ADVICE r4
                                              ((1 << 20) + 1, 40, False), (5_000_011, 24, True), (5_000_011, 45, False)])
def test_radix_sort_pairs_is_a_stable_sort(n, bits, with_vals):
    """p2w_sort_pairs_u64 == torch.sort(stable=True): keys of 0..63 significant bits (the pass count follows the data), many
    duplicates (stability decides their order), sizes around the 4096-key tile and on both sides of the two-level histogram scan's
    threshold, with given values and as an argsort."""
    from pointstowood_amd._lib import ptr, stream
    L = _L()
    g = torch.Generator().manual_seed(n + bits)
    if bits == 0:
        keys = torch.zeros(n, dtype=torch.int64)
    else:
        keys = torch.randint(0, 2 ** min(bits, 62), (n,), generator=g, dtype=torch.int64)
        if bits >= 63:
            keys = keys * 2 + torch.randint(0, 2, (n,), generator=g, dtype=torch.int64)
        keys[::3] = keys[0]                                             # long runs of equal keys
    keys = keys.cuda()
    vals = torch.randint(-5, 1 << 30, (n,), generator=g, dtype=torch.int32).cuda() if with_vals else None
    ko = torch.full_like(keys, -1)
    vo = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(L.p2w_sort_pairs_u64_ws_bytes(n)), dtype=torch.uint8, device="cuda")
    assert L.p2w_sort_pairs_u64(ptr(keys), ptr(ko), ptr(vals), ptr(vo), n, ptr(ws), ws.numel(), stream()) == 0
    ref_k, ref_i = torch.sort(keys, stable=True)
    assert torch.equal(ko, ref_k)
    assert torch.equal(vo.long(), ref_i if vals is None else vals[ref_i].long())
    assert L.p2w_sort_pairs_u64(ptr(keys), ptr(ko), ptr(vals), ptr(vo), n, ptr(ws), 16, stream()) == -4      # workspace
    assert L.p2w_sort_pairs_u64(ptr(keys), ptr(keys), ptr(vals), ptr(vo), n, ptr(ws), ws.numel(), stream()) == -1   # in place


@pytest.mark.parametrize("n,D", [(5000, 3), (20000, 6), (1, 4), (777, 16)])
def test_cells_nd_equals_voxel_grid_over_all_columns(n, D):
    """p2w_cells_nd == PyG voxel_grid(P, size) with batch = None over every column (oracle/ops.py), bit for bit."""
    from pointstowood_amd._lib import ptr, stream
    L = _L()
    g = torch.Generator().manual_seed(n + D)
    P = torch.rand(n, D, generator=g) * torch.linspace(3.0, 40.0, D)[None] - 7.0
    ref = O.voxel_grid(P, 2.0)
    dP = P.cuda()
    cell = torch.empty(n, dtype=torch.int64, device="cuda")
    ws = torch.empty(256, dtype=torch.uint8, device="cuda")
    assert L.p2w_cells_nd(ptr(dP), n, D, D, 2.0, ptr(cell), ptr(ws), ws.numel(), stream()) == 0
    assert torch.equal(cell.cpu(), ref)
    assert L.p2w_cells_nd(ptr(dP), n, 17, 17, 2.0, ptr(cell), ptr(ws), ws.numel(), stream()) == -1


@pytest.mark.parametrize("n,min_count", [(10000, 1), (10000, 40), (50000, 300), (5, 2), (4097, 4097)])
def test_key_runs_filter(n, min_count):
    from pointstowood_amd._lib import ptr, stream
    L = _L()
    g = torch.Generator().manual_seed(n)
    keys = torch.sort(torch.randint(0, max(2, n // 100), (n,), generator=g, dtype=torch.int64)).values
    if min_count == n:
        keys[:] = 3
    _, counts = torch.unique_consecutive(keys, return_counts=True)
    starts = torch.cumsum(counts, 0) - counts
    keep = counts >= min_count
    dk = keys.cuda()
    so, co = torch.full((n,), -1, dtype=torch.int32, device="cuda"), torch.full((n,), -1, dtype=torch.int32, device="cuda")
    no = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(L.p2w_key_runs_ws_bytes(n)), dtype=torch.uint8, device="cuda")
    assert L.p2w_key_runs(ptr(dk), n, min_count, ptr(so), ptr(co), ptr(no), ptr(ws), ws.numel(), stream()) == 0
    k = int(no)
    assert k == int(keep.sum())
    assert torch.equal(so[:k].cpu().long(), starts[keep]) and torch.equal(co[:k].cpu().long(), counts[keep])


def _host_voxelise(pc, *a, **kw):
    """The voxeliser's host-side glue over the oracle's tensor restatement of the grid step, on host tensors."""
    from oracle import preprocess as OP
    from pointstowood_amd import preprocessing as PP
    old, PP.backend = PP.backend, OP.TensorBackend
    try:
        return PP.voxelise(pc, *a, **kw)
    finally:
        PP.backend = old


@pytest.mark.parametrize("refl,mode", [(True, "compat"), (False, "compat"), (True, "xyz")])
def test_voxelise_on_the_gpu_equals_the_tensor_path(refl, mode):
    """preprocessing.voxelise on GPU tensors (cell ids, radix argsort and run filter by the HIP kernels) gives the voxels of the
    tensor-operation path on host tensors: same voxels, same order, same rows."""
    from pointstowood_amd import preprocessing as PP
    from tests.test_host_cpu import _plot
    pc = _plot(n=120000, seed=5, refl=refl)
    ref, nz_ref = _host_voxelise(pc, (2.0, 4.0), min_pts=64, max_pts=100000, mode=mode)
    got, nz = PP.voxelise(pc.cuda(), (2.0, 4.0), min_pts=64, max_pts=100000, mode=mode)
    assert len(got) == len(ref) and len(got) > 30
    assert (nz.cpu() - nz_ref).abs().max() <= 1e-5
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        assert torch.equal(a[:, :3].cpu(), b[:, :3])                          # membership and order: exact
        d = (a.cpu() - b).abs()
        assert d[:, 4:].max() <= 2e-5 and d[:, 3].max() <= 5e-4              # erfinv on GPU vs CPU: last bits of the quantiles


def test_voxeliser_keeps_non_finite_rows_out_of_every_voxel():
    """rows with a NaN / inf value take no part in the grid (minima, maxima, cell counts) and belong to no voxel, on the
    HIP path and on the tensor path alike (P2W_CELL_NONFINITE: they sort last as a run of their own, which is dropped); the
    voxels of the finite rows are those of the same cloud without the bad rows."""
    from oracle import preprocess as OP
    from pointstowood_amd import preprocessing as PP
    from pointstowood_amd._lib import ptr, stream
    from pointstowood_amd import synthetic_voxels as synth
    L = _L()
    pc = synth.forest_plot(60000, seed=4, side=12.0)
    n = pc.shape[0]
    g = torch.Generator().manual_seed(9)
    bad = torch.randperm(n, generator=g)[:300]
    dirty = pc.clone()
    dirty[bad[:100], 0] = float("nan")
    dirty[bad[100:200], 2] = float("inf")
    dirty[bad[200:], 3] = float("-inf")
    dirty[bad[:50], 1] = 1e30                           # (a dropped ROW's finite values must not stretch the grid either)
    # cell ids: the finite rows get the ids of the clean cloud restricted to them (same minima / maxima: the extremes are finite rows)
    keep = torch.ones(n, dtype=torch.bool)
    keep[bad] = False
    P = dirty[:, :4].contiguous()
    cell = torch.empty(n, dtype=torch.int64, device="cuda")
    ws = torch.empty(256, dtype=torch.uint8, device="cuda")
    assert L.p2w_cells_nd(ptr(P.cuda()), n, 4, 4, 2.0, ptr(cell), ptr(ws), ws.numel(), stream()) == 0
    ref = OP.cells_nd(P, 2.0)
    assert torch.equal(cell.cpu(), ref)
    assert bool((cell.cpu()[bad] == PP.CELL_NONFINITE).all()) and bool((cell.cpu()[keep] != PP.CELL_NONFINITE).all())
    assert torch.equal(ref[keep], OP.cells_nd(P[keep], 2.0))
    # whole voxeliser, ground normalisation off (its bucketize would see the NaN): GPU == host, no voxel holds a bad row
    dirty5 = torch.cat([dirty, torch.zeros(n, 1)], 1)
    vg, _ = PP.voxelise(dirty5.cuda(), (2.0, 4.0), 64, 16384, ground=False)
    vh, _ = _host_voxelise(dirty5, (2.0, 4.0), 64, 16384, ground=False)
    assert len(vg) == len(vh) > 10
    for a, b in zip(vg, vh):   # same rows in the same order (the normalised reflectance differs in erfinv's last bits between devices)
        assert a.shape == b.shape and torch.equal(a.cpu()[:, :3], b[:, :3]) and torch.allclose(a.cpu(), b, rtol=0, atol=5e-4)
        assert bool(torch.isfinite(b).all()) and bool(torch.isfinite(a).all())
    assert sum(v.shape[0] for v in vh) <= 2 * int(keep.sum())          # (two grid sizes; no bad row anywhere)
