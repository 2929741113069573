#!/usr/bin/env python3
"""Randomised stress of the grid-indexed searches against the whole-voxel (brute-force) kernels: random mixtures of
uniform / clustered / planar / duplicated points, random voxel counts and sizes, k, cell size, BOX on/off.  Where the
batch's cell grid fits a table, the table sampler must reproduce the sort sampler and the searches are repeated through the
INDEXED entry points (cell -> position tables instead of bisections: the engine's path).
Every result must be bit-identical.   python tools/stress_search.py [n_cases] [seed]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd._lib import SEARCH_BOX, SEARCH_Q_ROW_IN_W, SEARCH_X_INDEX_IN_W, lib, ptr, stream

L = lib()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
i32 = dict(dtype=torch.int32, device="cuda")


def cloud(n):
    kind = rng.integers(0, 5)
    ext = rng.choice([0.3, 2.0, 4.0, 15.0])
    p = rng.random((n, 3)) * ext
    if kind == 1:      # tight clusters far apart
        c = rng.random((rng.integers(1, 6), 3)) * ext
        p = c[rng.integers(0, len(c), n)] + rng.standard_normal((n, 3)) * rng.choice([1e-3, 0.02, 0.2])
    elif kind == 2:    # a plane / a line
        p[:, rng.integers(0, 3)] = rng.random() * ext
        if rng.random() < 0.3:
            p[:, rng.integers(0, 3)] = rng.random() * ext
    elif kind == 3:    # many exact duplicates
        p = p[rng.integers(0, max(1, n // 4), n)]
    elif kind == 4:    # lattice (massive distance ties)
        p = np.round(p / 0.05) * 0.05
    return (p + rng.choice([0.0, 100.0, -3.0])).astype(np.float32)


bad = 0
for case in range(n_cases):
    B = int(rng.integers(1, 5))
    sizes = [int(rng.integers(1, rng.choice([40, 600, 5000]))) for _ in range(B)]
    pos = np.concatenate([cloud(n) for n in sizes])
    batch = np.repeat(np.arange(B), sizes)
    res = float(rng.choice([0.01, 0.04, 0.08, 0.16, 0.5]))
    n = pos.shape[0]
    xyzr = torch.zeros((n, 4), dtype=torch.float32, device="cuda"); xyzr[:, :3] = torch.from_numpy(pos).cuda()
    csr = torch.zeros(B + 1, **i32); csr[1:] = torch.cumsum(torch.tensor(sizes), 0).int().cuda()
    idx, ptr_out, bo, order = (torch.empty(n, **i32), torch.empty(B + 1, **i32), torch.empty(n, **i32), torch.empty(n, **i32))
    skeys = torch.empty(n, dtype=torch.int64, device="cuda"); ckeys = torch.empty(n, dtype=torch.int64, device="cuda")
    grid = torch.zeros(8, dtype=torch.int64, device="cuda")
    ws = torch.empty(int(L.p2w_voxel_sample_ws_bytes(n)), dtype=torch.uint8, device="cuda")
    assert L.p2w_voxel_sample(ptr(xyzr), ptr(csr), B, n, res, ptr(idx), ptr(ptr_out), ptr(bo), ptr(order), ptr(skeys), ptr(ckeys),
                              ptr(grid), None, None, ptr(ws), ws.numel(), stream()) == 0
    m = int(ptr_out[B])
    rec = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    assert L.p2w_index_records(ptr(xyzr), ptr(order), ptr(csr), B, n, ptr(rec), stream()) == 0
    # the table sampler on the same batch (when its grid is small enough), with the cell -> position tables
    gdims = grid.cpu().numpy().view(np.uint8)[32:56].view(np.int64)
    cells = int(gdims[0] * gdims[1] * gdims[2]) * B
    tab = None
    if 0 < cells <= (1 << 22):
        cap = cells + int(rng.integers(0, 3)) * 1000
        wt = torch.empty(int(L.p2w_voxel_sample_table_ws_bytes(n, cap)), dtype=torch.uint8, device="cuda")
        t_idx, t_ptr, t_bo, t_order = (torch.empty(n, **i32), torch.empty(B + 1, **i32), torch.empty(n, **i32), torch.empty(n, **i32))
        t_sk = torch.empty(n, dtype=torch.int64, device="cuda"); t_ck = torch.empty(n, dtype=torch.int64, device="cuda")
        t_grid = torch.zeros(8, dtype=torch.int64, device="cuda")
        cs, css, status = torch.empty(cap + 1, **i32), torch.empty(cap + 1, **i32), torch.zeros(1, **i32)
        assert L.p2w_voxel_sample_table(ptr(xyzr), ptr(csr), B, n, res, ptr(t_idx), ptr(t_ptr), ptr(t_bo), ptr(t_order), ptr(t_sk),
                                        ptr(t_ck), ptr(t_grid), None, None, ptr(cs), ptr(css), ptr(status), cap, ptr(wt), wt.numel(),
                                        stream()) == 0
        if int(status) == 0:
            if not (torch.equal(t_ptr, ptr_out) and torch.equal(t_idx[:m], idx[:m]) and torch.equal(t_ck[:m], ckeys[:m])
                    and torch.equal(t_sk, skeys) and torch.equal(t_grid, grid)):
                bad += 1
                print(f"case {case}: table sampler != sort sampler sizes={sizes} res={res}")
            t_rec = torch.empty((n, 4), dtype=torch.float32, device="cuda")
            assert L.p2w_index_records(ptr(xyzr), ptr(t_order), ptr(csr), B, n, ptr(t_rec), stream()) == 0
            tab = (cs, css, t_rec, t_sk)
    coarse = xyzr[idx[:m].long()].contiguous()
    k = int(rng.choice([1, 2, 3, 8, 16, 32, 64]))
    box = int(rng.choice([0, SEARCH_BOX]))
    what = []
    # (a) other-level queries (fine points, cell order, row-in-w) over the coarse level
    out = []
    for g in (0, 1):
        nbr = torch.full((n, k), -7, **i32); deg = torch.full((n,), -7, **i32)
        if g:
            st = L.p2w_knn_grid(ptr(coarse), ptr(ckeys), ptr(ptr_out), ptr(grid), ptr(rec), None, ptr(csr), B, n, k, ptr(nbr),
                                ptr(deg), None, SEARCH_Q_ROW_IN_W | box, stream())
        else:
            st = L.p2w_knn(ptr(coarse), ptr(ptr_out), ptr(xyzr), None, ptr(csr), B, n, k, ptr(nbr), ptr(deg), None, 0, stream())
        assert st == 0
        out.append((nbr, deg))
    if not (torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])):
        what.append("knn(other level)")
    if tab is not None:
        nbr = torch.full((n, k), -7, **i32); deg = torch.full((n,), -7, **i32)
        assert L.p2w_knn_grid_indexed(ptr(coarse), ptr(ckeys), ptr(ptr_out), ptr(grid), ptr(tab[0]), ptr(tab[2]), None, ptr(csr), B, n, k,
                                      ptr(nbr), ptr(deg), None, SEARCH_Q_ROW_IN_W | box, stream()) == 0
        if not (torch.equal(out[0][0], nbr) and torch.equal(out[0][1], deg)):
            what.append("knn(other level, indexed)")
    # (b) subset queries (coarse points) over the fine level stored in cell order with index-in-w
    out = []
    for g in (0, 1):
        nbr = torch.full((m, k), -7, **i32); deg = torch.full((m,), -7, **i32)
        if g:
            st = L.p2w_knn_grid(ptr(rec), ptr(skeys), ptr(csr), ptr(grid), ptr(xyzr), ptr(idx), ptr(ptr_out), B, m, k, ptr(nbr),
                                ptr(deg), None, SEARCH_X_INDEX_IN_W | box, stream())
        else:
            st = L.p2w_knn(ptr(xyzr), ptr(csr), ptr(xyzr), ptr(idx), ptr(ptr_out), B, m, k, ptr(nbr), ptr(deg), None, 0, stream())
        assert st == 0
        out.append((nbr, deg))
    if not (torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])):
        what.append("knn(subset, index-in-w)")
    if tab is not None:   # candidates = the points in the TABLE sampler's cell-sorted order (inside a cell it may differ from the sort's)
        nbr = torch.full((m, k), -7, **i32); deg = torch.full((m,), -7, **i32)
        assert L.p2w_knn_grid_indexed(ptr(tab[2]), ptr(tab[3]), ptr(csr), ptr(grid), ptr(tab[1]), ptr(xyzr), ptr(idx), ptr(ptr_out), B, m, k,
                                      ptr(nbr), ptr(deg), None, SEARCH_X_INDEX_IN_W | box, stream()) == 0
        if not (torch.equal(out[0][0], nbr) and torch.equal(out[0][1], deg)):
            what.append("knn(subset, index-in-w, indexed)")
    # (c) ball query
    r = float(rng.choice([0.5, 1.0, 2.0, 3.5])) * res
    out = []
    for g in (0, 1):
        nbr = torch.full((m, k), -7, **i32); deg = torch.full((m,), -7, **i32)
        if g:
            st = L.p2w_ball_query_grid(ptr(rec), ptr(skeys), ptr(csr), ptr(grid), ptr(xyzr), ptr(idx), ptr(ptr_out), B, m, r, k,
                                       ptr(nbr), ptr(deg), SEARCH_X_INDEX_IN_W | box, stream())
        else:
            st = L.p2w_ball_query(ptr(xyzr), ptr(csr), ptr(xyzr), ptr(idx), ptr(ptr_out), B, m, r, k, ptr(nbr), ptr(deg), None, 0,
                                  stream())
        assert st == 0
        out.append((nbr, deg))
    if not (torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])):
        what.append("ball")
    if tab is not None:
        nbr = torch.full((m, k), -7, **i32); deg = torch.full((m,), -7, **i32)
        assert L.p2w_ball_query_grid_indexed(ptr(tab[2]), ptr(tab[3]), ptr(csr), ptr(grid), ptr(tab[1]), ptr(xyzr), ptr(idx), ptr(ptr_out), B,
                                             m, r, k, ptr(nbr), ptr(deg), SEARCH_X_INDEX_IN_W | box, stream()) == 0
        if not (torch.equal(out[0][0], nbr) and torch.equal(out[0][1], deg)):
            what.append("ball(indexed)")
    if what:
        bad += 1
        print(f"case {case}: MISMATCH {what} sizes={sizes} res={res} k={k} box={box} r={r}")
print(f"{n_cases} cases, {bad} mismatching")
sys.exit(1 if bad else 0)
