"""Operator-level parity on the GPU: every HIP operator (through the C ABI) against oracle/ops.py
on seeded inputs.  Index results must be bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import ops as O
from pointstowood_amd import synthetic_voxels as synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from pointstowood_amd import ops
    return ops


def _batch(sizes, side=2.0, seed=0, surface=False):
    vox = [(synth.surface_voxel if surface else synth.uniform_voxel)(side, n, seed + i, True) for i, n in enumerate(sizes)]
    return synth.collate(vox)


@pytest.mark.parametrize("sizes,res", [([2048], 0.04), ([3000, 17, 900], 0.08), ([16384, 512], 0.16), ([1], 0.04)])
def test_voxel_grid_and_cluster_exact(H, sizes, res):
    b = _batch(sizes, seed=3)
    cell_ref = O.voxel_grid(b["pos"], res, b["batch"])
    inv_ref, perm_ref = O.consecutive_cluster(cell_ref)
    cell = H.voxel_grid(b["pos"].cuda(), res, b["batch"].cuda())
    assert torch.equal(cell.cpu(), cell_ref)
    inv, perm = H.consecutive_cluster(cell)
    assert torch.equal(perm.cpu(), perm_ref)
    assert torch.equal(inv.cpu(), inv_ref)


@pytest.mark.parametrize("sizes,cap,surface", [([2048], 32, False), ([3000], 32, True), ([1500, 40, 700], 16, True),
                                               ([3000, 70], 100, True), ([2500], 65, True)])   # 65 .. 100: torch-cluster's upper range
def test_radius_exact(H, sizes, cap, surface):
    b = _batch(sizes, seed=5, surface=surface)
    idx = O.consecutive_cluster(O.voxel_grid(b["pos"], 0.04, b["batch"]))[1]
    ref = O.radius(b["pos"], b["pos"][idx], 0.08, b["batch"], b["batch"][idx], max_num_neighbors=cap)
    got = H.radius(b["pos"].cuda(), b["pos"][idx].cuda(), 0.08, b["batch"].cuda(), b["batch"][idx].cuda(),
                   max_num_neighbors=cap)
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize("sizes,k", [([2048], 32), ([2048], 16), ([700, 20, 3000], 32), ([5000], 2), ([9], 32), ([300], 64),
                                     ([1500, 80, 400], 100), ([90], 65)])   # 65 .. 100: torch-cluster's upper range (k <= 100)
def test_knn_exact(H, sizes, k):
    b = _batch(sizes, seed=7)
    idx = O.consecutive_cluster(O.voxel_grid(b["pos"], 0.08, b["batch"]))[1]
    ref = O.knn(b["pos"], b["pos"][idx], k, b["batch"], b["batch"][idx])
    got = H.knn(b["pos"].cuda(), b["pos"][idx].cuda(), k, b["batch"].cuda(), b["batch"][idx].cuda())
    assert torch.equal(got.cpu(), ref)


def test_knn_ties_prefer_lower_index(H):
    g = torch.Generator().manual_seed(1)
    base = torch.rand(400, 3, generator=g)
    x = torch.cat([base, base, base[:100]], 0)           # every distance occurs 2-3 times
    x = x[torch.randperm(x.shape[0], generator=g)]
    y = x[::7].clone()
    ref = O.knn(x, y, 32)
    got = H.knn(x.cuda(), y.cuda(), 32)
    assert torch.equal(got.cpu(), ref)
    # lattice: many exactly equal distances between distinct points
    lat = torch.stack(torch.meshgrid(*[torch.arange(9.0)] * 3, indexing="ij"), -1).reshape(-1, 3) * 0.125
    lat = lat[torch.randperm(lat.shape[0], generator=g)]
    assert torch.equal(H.knn(lat.cuda(), lat[:200].cuda(), 32).cpu(), O.knn(lat, lat[:200], 32))
    assert torch.equal(H.knn(lat.cuda(), lat[:200].cuda(), 100).cpu(), O.knn(lat, lat[:200], 100))   # the one-thread-per-query path
    with pytest.raises(RuntimeError):
        H.knn(lat.cuda(), lat[:200].cuda(), 101)
    assert torch.equal(H.radius(lat.cuda(), lat[:200].cuda(), 0.25, max_num_neighbors=32).cpu(),
                       O.radius(lat, lat[:200], 0.25, max_num_neighbors=32))


def test_knn_interpolate_and_pool(H):
    b = _batch([3000, 50, 1200], seed=9)
    idx = O.consecutive_cluster(O.voxel_grid(b["pos"], 0.16, b["batch"]))[1]
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(idx.numel(), 24, generator=g)
    ref = O.knn_interpolate(feat, b["pos"][idx], b["pos"], b["batch"][idx], b["batch"], k=2)
    got = H.knn_interpolate(feat.cuda(), b["pos"][idx].cuda(), b["pos"].cuda(), b["batch"][idx].cuda(), b["batch"].cuda(), k=2)
    assert (got.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()
    x = torch.randn(b["pos"].shape[0], 40, generator=g)
    assert torch.equal(H.global_max_pool(x.cuda(), b["batch"].cuda()).cpu(), O.global_max_pool(x, b["batch"]))


@pytest.mark.parametrize("M,N,K", [(1, 4, 4), (129, 8, 36), (1000, 192, 128), (4097, 512, 2048), (300, 640, 768), (77, 100, 516)])
def test_gemm_epilogue(M, N, K):
    """fp32 MFMA GEMM + fused epilogue vs fp64.  Tolerance: fp32 accumulation of K products."""
    import ctypes as C
    from pointstowood_amd import _lib
    from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream
    g = torch.Generator().manual_seed(M + N + K)
    lda = (K + 3) // 4 * 4
    A = torch.randn(M, lda, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    vec = lambda: torch.randn(N, generator=g)
    bias, s0, t0, s1, t1 = vec(), vec(), vec(), vec(), vec()
    R = torch.randn(M, N, generator=g)
    Np, Kp = _lib.packed_dims(N, K)
    Wp = torch.zeros(Np, Kp)
    Wp[:N, :K] = W
    d = lambda t: t.cuda().contiguous()
    dA, dW, db, ds0, dt0, ds1, dt1, dR = map(d, (A, Wp, bias, s0, t0, s1, t1, R))
    out = torch.full((M, N), float("nan"), device="cuda")
    ep = Epilogue(ptr(db), ptr(ds0), ptr(dt0), ptr(ds1), ptr(dt1), ptr(dR), N, 1, 1, 1, 1)
    check(lib().p2w_gemm(ptr(dA), lda, ptr(dW), M, N, K, C.byref(ep), ptr(out), N, stream()))
    v = A[:, :K].double() @ W.double().t() + bias.double()
    v = torch.relu(v) * s0.double() + t0.double()
    v = torch.relu(v) * s1.double() + t1.double()
    v = torch.relu(torch.relu(v) + R.double())
    err = (out.cpu().double() - v).abs().max().item()
    assert err <= 2e-5 * max(1.0, v.abs().max().item()), err
    # plain (no epilogue) into a wider output
    out2 = torch.zeros((M, N + 4), device="cuda")
    check(lib().p2w_gemm(ptr(dA), lda, ptr(dW), M, N, K, None, ptr(out2), N + 4, stream()))
    v2 = A[:, :K].double() @ W.double().t()
    assert (out2[:, :N].cpu().double() - v2).abs().max().item() <= 2e-5 * max(1.0, v2.abs().max().item())
    assert float(out2[:, N:].abs().max()) == 0.0


def _pack_h(W, prec):
    """Host-side H weights of W [N, K] for precision code `prec` (0 f16x3, 1 fp16, 2 bf16): (tensor on the GPU, 2^-e, K_pad)."""
    from pointstowood_amd import _lib
    N, K = W.shape
    Np, Kp = _lib.packed_dims(N, K, prec)
    Wp = torch.zeros(Np, Kp, dtype=torch.float64)
    Wp[:N, :K] = W.double()
    e = int(np.floor(np.log2(1024.0 / float(Wp.abs().max()))))
    Ws = Wp * 2.0 ** e
    if prec == 0:     # H layout: blocks of 32 k stored as [hi(32) | lo(32)]
        hi = Ws.float().half()
        lo = (Ws - hi.double()).float().half()
        w = torch.stack([hi.view(Np, Kp // 32, 32), lo.view(Np, Kp // 32, 32)], dim=2).reshape(Np, 2 * Kp)
    else:
        w = Ws.float().to(torch.float16 if prec == 1 else torch.bfloat16)
    return w.contiguous().cuda(), 2.0 ** -e, Kp


def _to_h(x, prec, ldh):
    """H form of an fp32 [m, F] tensor, produced ON THE DEVICE by p2w_concat_xyz_h2 with zero positions (so the
    kernels' own fp32 -> 16-bit conversion is what gets tested)."""
    from pointstowood_amd._lib import check, lib, ptr, stream
    m, F = x.shape
    assert F % 4 == 0 and ldh >= F + 4
    planes = 2 if prec == 0 else 1
    out = torch.full((m, planes * ldh), float("nan"), dtype=torch.bfloat16 if prec == 2 else torch.float16, device="cuda")
    zeros = torch.zeros((m, 4), device="cuda")
    check(lib().p2w_concat_xyz_h2(prec, ptr(x.cuda().contiguous()), F, ptr(zeros), m, ptr(out), ldh, stream()))
    return out


def _from_h(t, prec, ldh):
    """Values [m, ldh] (float64, on the CPU) of an H tensor: f16x3 rows are blocks of 32 columns stored [hi(32) | lo(32)]."""
    v = t.cpu().double()
    if prec != 0:
        return v[:, :ldh]
    b = v.view(v.shape[0], ldh // 32, 2, 32)
    return (b[:, :, 0] + b[:, :, 1]).reshape(v.shape[0], ldh)


# relative error bound of a K-term dot product per precision (operand rounding 2^-22 / 2^-11 / 2^-8, fp32 accumulate)
H_TOL = {0: 4e-5, 1: 3e-3, 2: 2.5e-2}


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("M,N,K,flags", [(300, 200, 100, 0), (1000, 512, 516, 0), (700, 256, 64, 2), (257, 130, 36, 0),
                                         (513, 512, 1024, 2), (513, 512, 1024, 1 | 16), (64, 3, 512, 0),
                                         # P2W_GEMM_TILE_64 = 1 << 24: the 64 x 128 tile, three workgroups per CU
                                         (300, 200, 100, 1 << 24), (1000, 512, 516, 1 << 24), (257, 130, 36, 1 << 24), (5, 128, 64, 1 << 24),
                                         (150000, 512, 128, 1 << 24), (70001, 256, 32, 1 << 24)])
def test_gemm_h_epilogue(prec, M, N, K, flags):
    """p2w_gemm_h2 (f16x3 / fp16 / bf16 MFMA, H operands, both tile sizes and tile orders) + fused epilogue vs fp64,
    fp32 and H outputs."""
    import ctypes as C
    from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, (K + 3) // 4 * 4, generator=g)
    A[:, K:] = 0
    W = torch.randn(N, K, generator=g) / K ** 0.5
    vec = lambda: torch.randn(N, generator=g)
    bias, s0, t0, s1, t1 = vec(), vec(), vec(), vec(), vec()
    R = torch.randn(M, N, generator=g)
    dW, wscale, Kp = _pack_h(W, prec)
    ka = 32 if prec == 0 else 64
    ldh_a = (A.shape[1] + 4 + ka - 1) // ka * ka
    Ah = _to_h(A, prec, ldh_a)
    d = lambda t: t.cuda().contiguous()
    db, ds0, dt0, ds1, dt1, dR = map(d, (bias, s0, t0, s1, t1, R))
    ep = Epilogue(ptr(db), ptr(ds0), ptr(dt0), ptr(ds1), ptr(dt1), ptr(dR), N, 1, 1, 1, 1)
    out = torch.full((M, N), float("nan"), device="cuda")
    ldh_o = (N + ka - 1) // ka * ka
    planes = 2 if prec == 0 else 1
    outh = torch.full((M, planes * ldh_o), float("nan"), dtype=Ah.dtype, device="cuda")
    check(lib().p2w_gemm_h2(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(out), N, ptr(outh), ldh_o, flags,
                            stream()))
    v = A[:, :K].double() @ W.double().t() + bias.double()
    v = torch.relu(v) * s0.double() + t0.double()
    v = torch.relu(v) * s1.double() + t1.double()
    v = torch.relu(torch.relu(v) + R.double())
    scale = max(1.0, v.abs().max().item())
    err = (out.cpu().double() - v).abs().max().item()
    assert err <= H_TOL[prec] * scale, err
    hv = _from_h(outh, prec, ldh_o)
    got_h = hv[:, :N]
    assert (got_h - out.cpu().double()).abs().max().item() <= (2e-6 if prec == 0 else 1e-3 if prec == 1 else 8e-3) * scale
    assert float(hv[:, N:ldh_o].abs().max() if ldh_o > N else 0.0) == 0.0          # pad columns of an H row are zero
    # plain (no epilogue), fp32 output only, into a wider output
    out2 = torch.zeros((M, N + 4), device="cuda")
    check(lib().p2w_gemm_h2(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, None, ptr(out2), N + 4, None, 0, flags, stream()))
    v2 = A[:, :K].double() @ W.double().t()
    assert (out2[:, :N].cpu().double() - v2).abs().max().item() <= H_TOL[prec] * max(1.0, v2.abs().max().item())
    assert float(out2[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("M,N,K,flags", [(300, 200, 100, 64), (1000, 512, 516, 64), (2100, 130, 1024, 64), (5, 512, 2048, 64),
                                         (70000, 512, 512, 64), (70000, 512, 512, 64 | 2), (17506, 2048, 2048, 0), (1122, 2048, 2048, 0),
                                         (33000, 256, 1024, 64 | 2)])
def test_gemm_h_stream_k_tail(prec, M, N, K, flags):
    """p2w_gemm_h2_sk: the rows behind the whole chip rounds as a stream-K tail (forced with P2W_GEMM_STREAMK = 64, or the
    library's own choice on the two level-3 shapes of the bench batch) - full epilogue, H residual in f16x3, fp32 + H outputs
    into wider rows - against fp64 and against p2w_gemm_h2; twice with the same bits (the fix-up adds the pieces in a fixed order)."""
    import ctypes as C
    from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream
    L = lib()
    g = torch.Generator().manual_seed(M + N + K)
    Kc = (K + 3) // 4 * 4
    A = torch.randn(M, Kc, generator=g)
    A[:, K:] = 0
    W = torch.randn(N, K, generator=g) / K ** 0.5
    vec = lambda: torch.randn(N, generator=g)
    bias, s0, t0 = vec(), vec(), vec()
    Rf = torch.randn(M, (N + 3) // 4 * 4, generator=g)
    Rf[:, N:] = 0
    dW, wscale, Kp = _pack_h(W, prec)
    ka = 32 if prec == 0 else 64
    planes = 2 if prec == 0 else 1
    ldh_a = (Kc + 4 + ka - 1) // ka * ka
    Ah = _to_h(A, prec, ldh_a)
    ldr = (Rf.shape[1] + 4 + ka - 1) // ka * ka
    Rh = _to_h(Rf, prec, ldr)
    R = _from_h(Rh, prec, ldr)[:, :N]
    db, ds0, dt0 = bias.cuda(), s0.cuda(), t0.cuda()
    ep = Epilogue(ptr(db), ptr(ds0), ptr(dt0), None, None, ptr(Rh), ldr, 1, 0, 0, 1)
    ldo, ldh_o = N + 6, (N + ka - 1) // ka * ka + ka
    ws = torch.empty(int(L.p2w_gemm_h2_sk_ws_bytes()), dtype=torch.uint8, device="cuda")

    def run(sk):
        out = torch.zeros((M, ldo), device="cuda")
        outh = torch.zeros((M, planes * ldh_o), dtype=Ah.dtype, device="cuda")
        if sk:
            check(L.p2w_gemm_h2_sk(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(out), ldo, ptr(outh), ldh_o,
                                   ptr(ws), ws.numel(), flags | 32, stream()))
        else:
            check(L.p2w_gemm_h2(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(out), ldo, ptr(outh), ldh_o,
                                (flags & ~64) | 32, stream()))
        return out, outh
    o1, h1 = run(True)
    o2, h2 = run(True)
    o0, h0 = run(False)
    assert torch.equal(o1, o2) and torch.equal(h1, h2)
    v = torch.relu(A[:, :K].double() @ W.double().t() + bias.double()) * s0.double() + t0.double()
    v = torch.relu(v + R)
    scale = max(1.0, v.abs().max().item())
    assert (o1[:, :N].cpu().double() - v).abs().max().item() <= H_TOL[prec] * scale
    assert float(o1[:, N:].abs().max()) == 0.0                      # nothing written beyond the N output columns
    assert (o1 - o0).abs().max().item() <= 4e-6 * scale * (K / 512) ** 0.5 + 1e-7   # the K range summed in pieces: last fp32 bits only
    hv = _from_h(h1, prec, ldh_o)
    assert (hv[:, :N] - o1[:, :N].cpu().double()).abs().max().item() <= (2e-6 if prec == 0 else 1e-3 if prec == 1 else 8e-3) * scale
    pad_to = (N + ka - 1) // ka * ka
    assert float(hv[:, N:pad_to].abs().max() if pad_to > N else 0.0) == 0.0
    assert float(hv[:, pad_to:].abs().max()) == 0.0                 # columns beyond the launch's own stay untouched


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,Mx,N,K,kw,sk", [(3000, 700, 640, 256, 2, 0), (81683, 17506, 640, 256, 2, 0), (70000, 20000, 512, 128, 2, 1),
                                            (17506, 9, 768, 512, 1, 1), (517, 40, 132, 32, 2, 0), (2100, 300, 132, 1024, 2, 1)])
def test_gemm_h_interpolated_residual(prec, M, Mx, N, K, kw, sk):
    """An FP module's layer 0 by linearity (model.py:149-153): relu(W_s skip + b + knn_interpolate(Z)) with the interpolation of
    the coarse level's rows Z done in the GEMM's epilogue (p2w_epilogue.interp, records from p2w_interp_weights) == the same GEMM
    with the interpolated rows given as an fp32 residual (p2w_interp_concat) to fp32 rounding, and == fp64; interior and edge
    tiles, one and two neighbours, rows with fewer neighbours than kw, plain launches and the split-K planner's row ranges."""
    import ctypes as C
    from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream
    L = lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    Z = torch.randn(Mx, N, generator=g)
    pc, pf = torch.rand(Mx, 4, generator=g), torch.rand(M, 4, generator=g)
    nbr = torch.randint(0, Mx, (M, kw), generator=g, dtype=torch.int32)
    deg = torch.full((M,), kw, dtype=torch.int32)
    deg[::7] = 1                                                            # fewer neighbours than kw
    pf[::11, :3] = pc[nbr[::11, 0].long(), :3]                              # a fine point ON its first neighbour (w = 1e16)
    dW, wscale, Kp = _pack_h(W, prec)
    ka, planes = (32, 2) if prec == 0 else (64, 1)
    ldh_a = (K + 4 + ka - 1) // ka * ka
    Ah = _to_h(A, prec, ldh_a)
    dZ, dpc, dpf, dn, dd, db = Z.cuda(), pc.cuda(), pf.cuda(), nbr.cuda(), deg.cuda(), bias.cuda()
    rec = torch.empty((M, 4), dtype=torch.int32, device="cuda")
    check(L.p2w_interp_weights(ptr(dpc), ptr(dpf), ptr(dn), ptr(dd), kw, M, ptr(rec), stream()))
    r = rec.cpu()
    a = r[:, 2:].contiguous().view(torch.float32)
    assert torch.equal(r[:, 0], nbr[:, 0]) and bool((r[:, 1] == torch.where(deg >= 2, nbr[:, kw - 1], nbr[:, 0])).all())
    assert float((a.sum(1) - 1).abs().max()) <= 2e-7 and bool((a[deg < 2, 1] == 0).all())
    R = torch.empty((M, N), device="cuda")
    check(L.p2w_interp_concat(ptr(dZ), N, ptr(dpc), ptr(dpf), ptr(dn), ptr(dd), kw, None, 0, M, ptr(R), N, stream()))
    ldh_o = (N + ka - 1) // ka * ka
    ws = torch.empty(int(L.p2w_gemm_h2_sk_ws_bytes()), dtype=torch.uint8, device="cuda")

    def run(ep, flags=0):
        outh = torch.zeros((M, planes * ldh_o), dtype=Ah.dtype, device="cuda")
        if sk:
            check(L.p2w_gemm_h2_sk(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), None, 0, ptr(outh), ldh_o, ptr(ws),
                                   ws.numel(), flags, stream()))
        else:
            check(L.p2w_gemm_h2(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), None, 0, ptr(outh), ldh_o, flags, stream()))
        return _from_h(outh, prec, ldh_o)
    fused = run(Epilogue(ptr(db), None, None, None, None, ptr(dZ), N, 0, 0, 0, 1, None, ptr(rec), Mx))
    plain = run(Epilogue(ptr(db), None, None, None, None, ptr(R), N, 0, 0, 0, 1))
    generic = run(Epilogue(ptr(db), None, None, None, None, ptr(dZ), N, 0, 0, 0, 1, None, ptr(rec), Mx), flags=4)
    Ad = _from_h(Ah, prec, ldh_a)[:, :K]
    v = torch.relu(Ad @ W.double().t() + bias.double() + R.cpu().double())
    scale = max(1.0, v.abs().max().item())
    tol = H_TOL[prec] * scale + (0 if prec == 0 else 2e-3 * scale)          # (H output of one plane: its own rounding)
    assert (fused[:, :N] - v).abs().max().item() <= tol
    assert (fused - plain).abs().max().item() <= (4e-6 if prec == 0 else 2e-3) * scale
    assert (fused - generic).abs().max().item() <= (4e-6 if prec == 0 else 2e-3) * scale
    assert float(fused[:, N:].abs().max() if ldh_o > N else 0.0) == 0.0
    if sk:   # the split-K fix-up's epilogue interpolates too (P2W_GEMM_STREAMK = 64 forces a split tail)
        forced = run(Epilogue(ptr(db), None, None, None, None, ptr(dZ), N, 0, 0, 0, 1, None, ptr(rec), Mx), flags=64)
        assert (forced - fused).abs().max().item() <= (4e-6 * (K / 512) ** 0.5 + 1e-7 if prec == 0 else 2e-3) * scale
    # what the boundary refuses: an H residual, records without rows (a forced 256 x 256 tile is ignored)
    ep = Epilogue(ptr(db), None, None, None, None, ptr(dZ), N, 0, 0, 0, 1, None, ptr(rec), Mx)
    outh = torch.zeros((M, planes * ldh_o), dtype=Ah.dtype, device="cuda")
    assert L.p2w_gemm_h2(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), None, 0, ptr(outh), ldh_o, 32, stream()) == -5
    assert torch.equal(run(ep, flags=2), fused)
    o32 = torch.empty((M, N), device="cuda")          # the fp32 engine does not interpolate: refused, never a read beyond Z's rows
    assert L.p2w_gemm(ptr(R), N, ptr(R), M, N, 4, C.byref(ep), ptr(o32), N, stream()) == -5
    ep0 = Epilogue(ptr(db), None, None, None, None, ptr(dZ), N, 0, 0, 0, 1, None, ptr(rec), 0)
    assert L.p2w_gemm_h2(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep0), None, 0, ptr(outh), ldh_o, 0, stream()) == -1


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("M,N,K,flags", [(1000, 512, 512, 0), (5, 512, 512, 0), (777, 100, 36, 0), (2049, 512, 64, 2), (2049, 512, 64, 1),
                                         (300, 130, 200, 0), (4096, 256, 128, 2), (32768, 512, 512, 0), (70000, 512, 512, 0),
                                         (70000, 512, 512, 1)])
def test_gemm_h_rowdot_head(prec, M, N, K, flags):
    """p2w_gemm_h2_rowdot (conv1 + BN + ReLU + conv2 for one class, model.py:241-243, without the [M, N] intermediate) against
    fp64, and against p2w_gemm_h2 + p2w_rowdot on the same operands; interior tiles (specialised epilogue), edge tiles and odd
    N (guarded epilogue), both tile sizes; deterministic."""
    import ctypes as C
    from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream
    g = torch.Generator().manual_seed(3 * M + N + K)
    A = torch.randn(M, (K + 3) // 4 * 4, generator=g)
    A[:, K:] = 0
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias, dotw = torch.randn(N, generator=g), torch.randn(N, generator=g)
    dW, wscale, Kp = _pack_h(W, prec)
    ka = 32 if prec == 0 else 64
    ldh_a = (A.shape[1] + 4 + ka - 1) // ka * ka
    Ah = _to_h(A, prec, ldh_a)
    db, dw = bias.cuda(), dotw.cuda()
    ep = Epilogue(ptr(db), None, None, None, None, None, 0, 1, 0, 0, 0)          # the head: bias (BN folded) + ReLU
    need = int(lib().p2w_gemm_h2_rowdot_ws_bytes(M, N))
    assert need >= ((N + 63) // 64) * M * 4
    ws = torch.full((need,), 0xFF, dtype=torch.uint8, device="cuda")
    out = torch.full((M,), float("nan"), device="cuda")
    args = (prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(dw), 0.25, ptr(out), ptr(ws), need, flags, stream())
    check(lib().p2w_gemm_h2_rowdot(*args))
    v = torch.relu(A[:, :K].double() @ W.double().t() + bias.double()) @ dotw.double() + 0.25
    scale = max(1.0, float((torch.relu(A[:, :K].double() @ W.double().t() + bias.double()).abs() @ dotw.double().abs()).max()))
    assert (out.cpu().double() - v).abs().max().item() <= H_TOL[prec] * scale
    # the unfused pair on the same operands differs only by fp32 summation order
    hd = torch.empty((M, N), device="cuda")
    check(lib().p2w_gemm_h2(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(hd), N, None, 0, flags, stream()))
    ref = (hd.double() @ dw.double() + 0.25).cpu()
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-6 * scale
    out2 = torch.full((M,), float("nan"), device="cuda")
    ws.fill_(0x7F)
    check(lib().p2w_gemm_h2_rowdot(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(dw), 0.25, ptr(out2), ptr(ws),
                                   need, flags, stream()))
    assert torch.equal(out, out2)                                                  # fixed summation order
    assert lib().p2w_gemm_h2_rowdot(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(dw), 0.25, ptr(out2), ptr(ws),
                                    need - 1, flags, stream()) == -4            # P2W_EWORKSPACE
    epr = Epilogue(ptr(db), None, None, None, None, ptr(hd), N, 1, 0, 0, 0)
    assert lib().p2w_gemm_h2_rowdot(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(epr), ptr(dw), 0.25, ptr(out2), ptr(ws),
                                    need, flags, stream()) == -5               # residual: P2W_EUNSUPPORTED


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("M,N,K,flags", [(1000, 128, 512, 0), (515, 96, 100, 0), (2048, 256, 64, 2), (300, 32, 40, 0),
                                         (90000, 512, 128, 1 << 24), (515, 96, 100, 1 << 24)])
def test_gemm_h_writes_into_wider_rows_and_reads_an_h_residual(prec, M, N, K, flags):
    """The two boundary features the in-place skip concatenation uses (engine.py): (1) ldh_o is a row PITCH - the launch writes
    its N columns (+ zero pad to the K-slab boundary) at a column offset of wider rows and leaves every other column alone;
    (2) P2W_GEMM_RESIDUAL_H - the residual is read from an H tensor.  Against fp64 and against the fp32-residual launch."""
    import ctypes as C
    from pointstowood_amd._lib import GEMM_RESIDUAL_H, Epilogue, check, lib, ptr, stream
    g = torch.Generator().manual_seed(7 * M + N + K)
    A = torch.randn(M, (K + 3) // 4 * 4, generator=g)
    A[:, K:] = 0
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    R = torch.randn(M, (N + 3) // 4 * 4, generator=g)
    R[:, N:] = 0
    dW, wscale, Kp = _pack_h(W, prec)
    ka, planes = (32, 2) if prec == 0 else (64, 1)
    ldh_a = (A.shape[1] + 4 + ka - 1) // ka * ka
    ldr = (R.shape[1] + 4 + ka - 1) // ka * ka
    Ah, Rh = _to_h(A, prec, ldh_a), _to_h(R, prec, ldr)
    Rv = _from_h(Rh, prec, ldr)[:, :N]                       # the residual values the kernel sees (exactly, for every precision)
    db = bias.cuda()
    off, pitch = 2 * ka, 2 * ka + (N + ka - 1) // ka * ka + ka      # two slabs of foreign columns in front, one behind
    wide = torch.full((M, planes * pitch), 7.0, dtype=Ah.dtype, device="cuda")
    view = wide[:, planes * off:]
    ep = Epilogue(ptr(db), None, None, None, None, ptr(Rh), ldr, 0, 0, 0, 1)
    out = torch.full((M, N), float("nan"), device="cuda")
    check(lib().p2w_gemm_h2(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(out), N, ptr(view), pitch,
                            flags | GEMM_RESIDUAL_H, stream()))
    v = torch.relu(A[:, :K].double() @ W.double().t() + bias.double() + Rv)
    scale = max(1.0, v.abs().max().item())
    assert (out.cpu().double() - v).abs().max().item() <= H_TOL[prec] * scale
    got = _from_h(wide, prec, pitch)
    npad = (N + ka - 1) // ka * ka
    assert (got[:, off:off + N] - out.cpu().double()).abs().max().item() <= (2e-6 if prec == 0 else 1e-3 if prec == 1 else 8e-3) * scale
    assert float(got[:, off + N:off + npad].abs().max() if npad > N else 0.0) == 0.0     # the launch's own zero pad
    foreign = torch.cat([wide[:, : planes * off], wide[:, planes * (off + npad):]], 1)
    assert bool((foreign == 7.0).all())                                                  # nobody else's columns were touched
    # the same launch with the residual given in fp32 (the values the H tensor holds): identical outputs
    dR = Rv.float().cuda().contiguous()
    ep2 = Epilogue(ptr(db), None, None, None, None, ptr(dR), N, 0, 0, 0, 1)
    out2 = torch.full((M, N), float("nan"), device="cuda")
    check(lib().p2w_gemm_h2(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep2), ptr(out2), N, None, 0, flags, stream()))
    assert (out2 - out).abs().max().item() <= 2e-6 * scale


def test_interp_and_stem_write_only_their_columns_of_wider_rows():
    """p2w_interp_concat_h2 with skip = NULL and p2w_stem_h2 treat ldh as the row pitch: the interpolated part / the stem's
    channels (+ zero pad to the slab boundary) are written, the other columns of the rows stay as they were."""
    from pointstowood_amd._lib import check, lib, ptr, stream
    b = _batch([1500, 300], seed=31)
    idx = O.consecutive_cluster(O.voxel_grid(b["pos"], 0.16, b["batch"]))[1]
    Fc, m = 64, b["pos"].shape[0]
    feat = torch.randn(idx.numel(), Fc, generator=torch.Generator().manual_seed(5))
    ref = O.knn_interpolate(feat, b["pos"][idx], b["pos"], b["batch"][idx], b["batch"], k=2)
    from pointstowood_amd import ops as H
    nbr, deg = H._search("knn", b["pos"][idx].cuda(), b["pos"].cuda(), None, b["batch"][idx].cuda(), b["batch"].cuda(), 2)
    rc, rf = H._xyzr(b["pos"][idx].cuda()), H._xyzr(b["pos"].cuda())
    for prec, (ka, planes, dt) in {0: (32, 2, torch.float16), 1: (64, 1, torch.float16), 2: (64, 1, torch.bfloat16)}.items():
        pitch = Fc + 2 * ka
        wide = torch.full((m, planes * pitch), 3.0, dtype=dt, device="cuda")
        dfeat = feat.cuda()
        check(lib().p2w_interp_concat_h2(prec, ptr(dfeat), Fc, ptr(rc), ptr(rf), ptr(nbr), ptr(deg), 2, None, 0, m,
                                         ptr(wide), pitch, stream()))
        got = _from_h(wide, prec, pitch)
        tol = {0: 2e-6, 1: 1e-3, 2: 8e-3}[prec] * float(ref.abs().max())
        assert (got[:, :Fc] - ref.double()).abs().max().item() <= tol + 1e-5 * float(ref.abs().max())
        assert bool((wide[:, planes * Fc:] == 3.0).all())
        # the stem into the last slab of the same rows
        Cw = 8
        w_, b_ = torch.randn(Cw, 3, generator=torch.Generator().manual_seed(6)), torch.randn(Cw, generator=torch.Generator().manual_seed(7))
        x0 = torch.empty((m, Cw), device="cuda")
        view = wide[:, planes * (Fc + ka):]
        dw_, db_ = w_.cuda(), b_.cuda()          # (kept alive: a temporary's memory is recycled by the next allocation)
        check(lib().p2w_stem_h2(prec, ptr(rf), m, ptr(dw_), ptr(db_), Cw, ptr(x0), ptr(view), pitch, None, stream()))
        sref = torch.relu(b["pos"].double() @ w_.double().t() + b_.double())
        assert (x0.cpu().double() - sref).abs().max().item() <= 1e-5
        got = _from_h(wide, prec, pitch)
        assert (got[:, Fc + ka:Fc + ka + Cw] - sref).abs().max().item() <= {0: 2e-6, 1: 2e-3, 2: 2e-2}[prec] * max(1.0, float(sref.abs().max()))
        assert float(got[:, Fc + ka + Cw:].abs().max()) == 0.0                         # zero pad up to the slab boundary = row end
        assert bool((wide[:, planes * Fc: planes * (Fc + ka)] == 3.0).all())          # the slab between them: untouched


def test_gemm_h_random_shapes():
    """80 random (shape, precision, epilogue, tile flag) cases of p2w_gemm_h2 against fp64 (tools/stress_gemm.py: edge tiles of both
    tile sizes, single-slab K, one-row / one-column problems, odd N through the guarded epilogue, every output combination)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_gemm.py"), "80", "5"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "80 cases ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_gemm_h_rejects_bad_arguments():
    import ctypes as C
    from pointstowood_amd._lib import lib, ptr, stream
    L = lib()
    A = torch.zeros((64, 128), dtype=torch.float16, device="cuda")
    W = torch.zeros((256, 128), dtype=torch.float16, device="cuda")
    o = torch.zeros((64, 8), device="cuda")
    args = lambda prec, ldh, flags: (prec, ptr(A), ldh, ptr(W), 1.0, 64, 8, 64, None, ptr(o), 8, None, 0, flags, stream())
    assert L.p2w_gemm_h2(*args(0, 64, 0)) == 0
    assert L.p2w_gemm_h2(*args(3, 64, 0)) == -1          # unknown precision
    assert L.p2w_gemm_h2(*args(0, 40, 0)) == -1          # A rows not padded to whole K slabs
    assert L.p2w_gemm_h2(*args(1, 32, 0)) == -1          # single-plane slabs are 64 wide
    assert L.p2w_gemm_h2(*args(0, 64, 3)) == -1          # both tile sizes forced
    assert L.p2w_gemm_h2(0, None, 64, ptr(W), 1.0, 64, 8, 64, None, ptr(o), 8, None, 0, 0, stream()) == -2
    a, b = C.c_int32(), C.c_int32()
    assert L.p2w_packed_dims_h(1, 100, 100, C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == (256, 128)
    assert L.p2w_packed_dims_h(0, 100, 100, C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == (256, 128)
    assert L.p2w_packed_dims_h(0, 100, 40, C.byref(a), C.byref(b)) == 0 and b.value == 64
    assert L.p2w_packed_dims_h(2, 100, 40, C.byref(a), C.byref(b)) == 0 and b.value == 64
    assert L.p2w_packed_dims_h(0, 100, 20, C.byref(a), C.byref(b)) == 0 and b.value == 32
    assert L.p2w_packed_dims_h(7, 1, 1, C.byref(a), C.byref(b)) == -1


def test_sa_conv_h_reports_its_limits():
    """The fused PointNetConv keeps two tables in LDS and uses 32-bit offsets: sizes beyond them are P2W_EUNSUPPORTED (-5),
    never a wrong answer (include/p2w.h)."""
    from pointstowood_amd._lib import lib, ptr, stream
    L = lib()
    dev = "cuda"
    M, n_src, C1 = 8, 8, 64
    z = lambda *sh, dt=torch.float32: torch.zeros(sh, dtype=dt, device=dev)
    xyzr, idx, bd, sf = z(n_src, 4), z(M, dt=torch.int32), z(M, dt=torch.int32), torch.ones(1, device=dev)
    nbr, deg, w1r4 = z(M, 32, dt=torch.int32), z(M, dt=torch.int32), z(4, 64)
    ws = torch.zeros(int(L.p2w_sa_conv_h_ws_bytes(M, 4)) + 65536, dtype=torch.uint8, device=dev)

    def call(C2, ldp=64, C1_=C1):
        P = z(n_src + 1, ldp)
        W2 = z(max(256, (C2 + 255) // 256 * 256), 2 * ((C1_ + 31) // 32 * 32), dt=torch.float16)
        v = z(C2)
        out = z(M, C2)
        return L.p2w_sa_conv_h(0, ptr(P), ldp, n_src, ptr(xyzr), ptr(idx), ptr(bd), ptr(sf), ptr(nbr), ptr(deg), 32, M, ptr(w1r4),
                               ptr(W2), 1.0, C1_, C2, ptr(v), ptr(v), ptr(v), ptr(out), C2, None, 0, ptr(ws), ws.numel(), 0, stream())
    assert call(256) == 0
    assert call(1024) == 0
    assert call(1028) == -5                         # per-column epilogue table: C2 <= 1024
    w1r4 = z(4, 576)
    assert call(256, ldp=576, C1_=544) == -5        # layer-1 geometry weights table: round_up(C1) <= 512
    torch.cuda.synchronize()


def test_c_abi_rejects_bad_arguments():
    from pointstowood_amd._lib import lib, ptr, stream
    x = torch.zeros(16, 4, device="cuda")
    p = torch.tensor([0, 16], dtype=torch.int32, device="cuda")
    n = torch.zeros(16, 101, dtype=torch.int32, device="cuda")
    d = torch.zeros(16, dtype=torch.int32, device="cuda")
    L = lib()
    assert L.p2w_knn(ptr(x), ptr(p), ptr(x), None, ptr(p), 1, 16, 101, ptr(n), ptr(d), None, 0, stream()) == -1   # k > 100 (P2W_MAX_K_WIDE)
    assert L.p2w_knn(ptr(x), ptr(p), ptr(x), None, ptr(p), 1, 16, 100, ptr(n), ptr(d), None, 0, stream()) == 0    # 65 .. 100: the wide path
    g8 = torch.zeros(8, dtype=torch.int64, device="cuda")
    k64 = torch.zeros(16, dtype=torch.int64, device="cuda")
    assert L.p2w_knn_grid(ptr(x), ptr(k64), ptr(p), ptr(g8), ptr(x), None, ptr(p), 1, 16, 65, ptr(n), ptr(d), None, 0, stream()) == -1   # grid searches: k <= 64
    assert L.p2w_knn(None, ptr(p), ptr(x), None, ptr(p), 1, 16, 8, ptr(n), ptr(d), None, 0, stream()) == -2       # NULL
    assert L.p2w_knn(x.data_ptr() + 4, ptr(p), ptr(x), None, ptr(p), 1, 16, 8, ptr(n), ptr(d), None, 0, stream()) == -3  # alignment
    assert b"NULL" in L.p2w_strerror(-2)


@pytest.mark.parametrize("prec", [0, 1])
def test_h_conversion_saturates_instead_of_overflowing(prec):
    """Activations beyond the fp16 range must degrade gracefully, never inf/NaN: the f16x3 hi plane saturates at 65504
    and lo carries the rest (usable to ~1.3e5); the single fp16 plane clamps at +-65504."""
    from pointstowood_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(3)
    M, K, N = 300, 64, 32
    A = (torch.rand(M, K, generator=g) * 2 - 1) * 1.2e5          # up to +-1.2e5 > 65504
    W = torch.randn(N, K, generator=g) / 8
    dW, wscale, Kp = _pack_h(W, prec)
    Ah = _to_h(A, prec, 128)
    assert bool(torch.isfinite(Ah.float()).all())
    out = torch.full((M, N), float("nan"), device="cuda")
    check(lib().p2w_gemm_h2(prec, ptr(Ah), 128, ptr(dW), wscale, M, N, K, None, ptr(out), N, None, 0, 0, stream()))
    assert bool(torch.isfinite(out).all())
    if prec == 0:
        ref = A.double() @ W.double().t()
        assert (out.cpu().double() - ref).abs().max() <= 2e-3 * ref.abs().max()
    else:
        ref = A.clamp(-65504, 65504).double() @ W.double().t()
        assert (out.cpu().double() - ref).abs().max() <= 5e-3 * ref.abs().max()


def _sorted_level(b, res):
    """Level records in the sampler's cell order via the C ABI: (xyzr, ptr, order, sorted records, tile boxes)."""
    from pointstowood_amd._lib import lib, ptr, stream
    L = lib()
    pos, batch = b["pos"].cuda(), b["batch"].cuda()
    n, B = pos.shape[0], int(batch.max()) + 1
    xyzr = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    xyzr[:, :3] = pos
    csr = torch.zeros(B + 1, dtype=torch.int32, device="cuda")
    csr[1:] = torch.cumsum(torch.bincount(batch, minlength=B), 0).int()
    i32 = dict(dtype=torch.int32, device="cuda")
    idx, ptr_out, batch_out, order = torch.empty(n, **i32), torch.empty(B + 1, **i32), torch.empty(n, **i32), torch.empty(n, **i32)
    ws = torch.empty(int(L.p2w_voxel_sample_ws_bytes(n)), dtype=torch.uint8, device="cuda")
    skeys = torch.empty(n, dtype=torch.int64, device="cuda")
    ckeys = torch.empty(n, dtype=torch.int64, device="cuda")
    grid = torch.zeros(8, dtype=torch.int64, device="cuda")
    assert L.p2w_voxel_sample(ptr(xyzr), ptr(csr), B, n, res, ptr(idx), ptr(ptr_out), ptr(batch_out), ptr(order), ptr(skeys),
                              ptr(ckeys), ptr(grid), None, None, ptr(ws), ws.numel(), stream()) == 0
    rec = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    assert L.p2w_index_records(ptr(xyzr), ptr(order), ptr(csr), B, n, ptr(rec), stream()) == 0
    box = torch.empty((L.p2w_tile_bbox_count(B, n), 6), dtype=torch.float32, device="cuda")
    assert L.p2w_tile_bbox(ptr(rec), ptr(csr), B, n, ptr(box), stream()) == 0
    m = int(ptr_out[B])
    return dict(L=L, xyzr=xyzr, csr=csr, B=B, n=n, order=order, rec=rec, box=box, idx=idx[:m], ptr_out=ptr_out, m=m,
                skeys=skeys, ckeys=ckeys[:m], grid=grid)


@pytest.mark.parametrize("sizes,res,surface", [([5000], 0.04, False), ([1500, 40, 2600], 0.08, True), ([16384, 3000, 1, 700], 0.04, False),
                                               ([9000, 0, 4000], 0.16, False)])
def test_table_sampler_equals_sort_sampler(sizes, res, surface):
    """p2w_voxel_sample_table (direct cell table, no sort) against p2w_voxel_sample on the same batch: representatives,
    CSR, batch, cell keys, grid and ranks identical; the sorted order is a permutation in ascending cell order (the
    order inside a cell is free).  A table that is too small must say so."""
    import ctypes as C
    from pointstowood_amd._lib import lib, ptr, stream
    vox = [synth.uniform_voxel(2.0, max(n, 1), 77 + i, True) if not surface else synth.surface_voxel(2.0, max(n, 1), 77 + i, True)
           for i, n in enumerate(sizes)]
    b = synth.collate([v for v, n in zip(vox, sizes)])
    keep = torch.cat([torch.ones(max(n, 1), dtype=torch.bool) if n > 0 else torch.zeros(1, dtype=torch.bool) for n in sizes])
    pos = b["pos"][keep].cuda()
    n, B = pos.shape[0], len(sizes)
    L = lib()
    xyzr = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    xyzr[:, :3] = pos
    csr = torch.zeros(B + 1, dtype=torch.int32)
    csr[1:] = torch.cumsum(torch.tensor(sizes), 0).int()
    csr = csr.cuda()
    i32 = dict(dtype=torch.int32, device="cuda")
    i64 = dict(dtype=torch.int64, device="cuda")

    def outputs():
        return dict(idx=torch.full((n,), -7, **i32), ptr=torch.full((B + 1,), -7, **i32), batch=torch.full((n,), -7, **i32),
                    order=torch.full((n,), -7, **i32), skeys=torch.full((n,), -7, **i64), ckeys=torch.full((n,), -7, **i64),
                    grid=torch.zeros(8, **i64), inv=torch.full((n,), -7, **i32), rsort=torch.full((n,), -7, **i32))
    a, t = outputs(), outputs()
    ws = torch.empty(int(L.p2w_voxel_sample_ws_bytes(n)), dtype=torch.uint8, device="cuda")
    assert L.p2w_voxel_sample(ptr(xyzr), ptr(csr), B, n, res, ptr(a["idx"]), ptr(a["ptr"]), ptr(a["batch"]), ptr(a["order"]),
                              ptr(a["skeys"]), ptr(a["ckeys"]), ptr(a["grid"]), ptr(a["inv"]), ptr(a["rsort"]), ptr(ws), ws.numel(),
                              stream()) == 0
    cells = B * (int(2.3 / res) + 3) ** 3
    wt = torch.empty(int(L.p2w_voxel_sample_table_ws_bytes(n, cells)), dtype=torch.uint8, device="cuda")
    status = torch.full((1,), 9, **i32)
    cs = torch.full((cells + 1,), -7, **i32)          # cell -> first representative at or after it
    css = torch.full((cells + 1,), -7, **i32)         # cell -> first point at or after it in the cell-sorted order
    assert L.p2w_voxel_sample_table(ptr(xyzr), ptr(csr), B, n, res, ptr(t["idx"]), ptr(t["ptr"]), ptr(t["batch"]), ptr(t["order"]),
                                    ptr(t["skeys"]), ptr(t["ckeys"]), ptr(t["grid"]), ptr(t["inv"]), ptr(t["rsort"]), ptr(cs), ptr(css),
                                    ptr(status), cells, ptr(wt), wt.numel(), stream()) == 0
    torch.cuda.synchronize()
    assert int(status) == 0
    m = int(a["ptr"][B])
    # the cell -> position tables are lower bounds of the sorted keys, for every key up to the grid's cell count
    dims = t["grid"].view(torch.int64)[4:7].tolist()
    b_used = int((a["ptr"][1:] > a["ptr"][:-1]).nonzero().max()) - int((a["ptr"][1:] > a["ptr"][:-1]).nonzero().min()) + 1
    T = dims[0] * dims[1] * dims[2] * b_used
    probe = torch.cat([torch.arange(0, T + 1, max(1, T // 5000), device="cuda"), torch.tensor([T], device="cuda")])
    assert torch.equal(cs[probe].long(), torch.searchsorted(a["ckeys"][:m].contiguous(), probe))
    assert torch.equal(css[probe].long(), torch.searchsorted(a["skeys"].contiguous(), probe))
    assert torch.equal(t["ptr"], a["ptr"]) and m > 0
    for k in ("idx", "batch", "ckeys"):
        assert torch.equal(t[k][:m], a[k][:m]), k
    assert torch.equal(t["grid"], a["grid"]) and torch.equal(t["inv"], a["inv"])
    assert torch.equal(t["skeys"], a["skeys"])                                   # the multiset of keys in ascending order
    assert torch.equal(torch.sort(t["order"].long()).values, torch.arange(n, device="cuda"))
    assert torch.equal(t["skeys"], torch.gather(a["skeys"], 0, torch.argsort(a["order"].long())[t["order"].long()]))   # each point keeps its key
    assert torch.equal(t["rsort"], t["inv"][t["order"].long()])
    # without the optional outputs
    t2 = outputs()
    assert L.p2w_voxel_sample_table(ptr(xyzr), ptr(csr), B, n, res, ptr(t2["idx"]), ptr(t2["ptr"]), ptr(t2["batch"]), None, None,
                                    ptr(t2["ckeys"]), None, ptr(t2["inv"]), None, None, None, ptr(status), cells, ptr(wt), wt.numel(), stream()) == 0
    assert torch.equal(t2["idx"][:m], a["idx"][:m]) and torch.equal(t2["inv"], a["inv"]) and int(status) == 0
    # a table that cannot hold the grid reports it instead of writing out of bounds
    small = 1000
    assert L.p2w_voxel_sample_table(ptr(xyzr), ptr(csr), B, n, res, ptr(t2["idx"]), ptr(t2["ptr"]), ptr(t2["batch"]), None, None,
                                    None, None, None, None, None, None, ptr(status), small, ptr(wt), wt.numel(), stream()) == 0
    assert int(status) == 1 and int(t2["ptr"].abs().max()) == 0          # ... and hands out an empty level
    assert L.p2w_voxel_sample_table(ptr(xyzr), ptr(csr), B, n, res, ptr(t2["idx"]), ptr(t2["ptr"]), ptr(t2["batch"]), None, None,
                                    None, None, None, None, None, None, ptr(status), cells, ptr(wt), 1024, stream()) == -4      # workspace


def test_prepared_table_sampler_keeps_its_workspace_state():
    """p2w_voxel_sample_table_prepared (6 launches: the engine's call) on a workspace prepared ONCE gives the results of
    p2w_voxel_sample_table call after call - two different batches alternating, an overflowing call (table too small) and an
    empty batch in between: every call leaves the between-calls state (bounding-box words, scan counter) in place again."""
    from pointstowood_amd._lib import lib, ptr, stream
    L = lib()
    i32, i64 = dict(dtype=torch.int32, device="cuda"), dict(dtype=torch.int64, device="cuda")
    res = 0.08

    def batch(sizes, seed):
        b = synth.collate([synth.uniform_voxel(2.0, n, seed + i, True) for i, n in enumerate(sizes)])
        xyzr = torch.zeros((b["pos"].shape[0], 4), device="cuda")
        xyzr[:, :3] = b["pos"].cuda()
        return xyzr, torch.tensor([0] + list(np.cumsum(sizes)), **i32), len(sizes)

    def run(fn, xyzr, csr, B, cells, wt, status):
        n = xyzr.shape[0]
        o = dict(idx=torch.full((n,), -7, **i32), ptr=torch.full((B + 1,), -7, **i32), batch=torch.full((n,), -7, **i32),
                 order=torch.full((n,), -7, **i32), skeys=torch.full((n,), -7, **i64), ckeys=torch.full((n,), -7, **i64),
                 grid=torch.zeros(8, **i64), inv=torch.full((n,), -7, **i32), rsort=torch.full((n,), -7, **i32),
                 cs=torch.full((cells + 1,), -7, **i32), css=torch.full((cells + 1,), -7, **i32))
        assert fn(ptr(xyzr), ptr(csr), B, n, res, ptr(o["idx"]), ptr(o["ptr"]), ptr(o["batch"]), ptr(o["order"]), ptr(o["skeys"]),
                  ptr(o["ckeys"]), ptr(o["grid"]), ptr(o["inv"]), ptr(o["rsort"]), ptr(o["cs"]), ptr(o["css"]), ptr(status), cells, ptr(wt),
                  wt.numel(), stream()) == 0
        torch.cuda.synchronize()
        return o, int(status)
    b1, b2 = batch([9000, 300, 4000], 11), batch([2500, 7000], 40)
    cells = 3 * (int(2.3 / res) + 3) ** 3
    nmax = max(b1[0].shape[0], b2[0].shape[0])
    need = int(L.p2w_voxel_sample_table_ws_bytes(nmax, cells))
    w_ref = torch.empty(need, dtype=torch.uint8, device="cuda")
    wt = torch.randint(0, 255, (need,), dtype=torch.uint8, device="cuda")        # a workspace somebody else has written
    assert L.p2w_voxel_sample_table_prepare(ptr(wt), wt.numel(), stream()) == 0
    status = torch.full((1,), 9, **i32)
    ref = {0: run(L.p2w_voxel_sample_table, *b1, cells, w_ref, status), 1: run(L.p2w_voxel_sample_table, *b2, cells, w_ref, status)}
    for step, which in enumerate([0, 1, 1, 0, "small", 0, "empty", 1, 0]):
        if which == "small":     # the grid does not fit: status 1, an empty level - and the state survives
            o, st = run(L.p2w_voxel_sample_table_prepared, *b1, 1000, wt, status)
            assert st == 1 and int(o["ptr"].abs().max()) == 0
            continue
        if which == "empty":     # no points at all (n_bound > 0 rows, every voxel empty)
            xyzr, _, B = b2
            o, st = run(L.p2w_voxel_sample_table_prepared, xyzr, torch.zeros(B + 1, **i32), B, cells, wt, status)
            assert st == 0 and int(o["ptr"].abs().max()) == 0
            continue
        (r, rst), (o, st) = ref[which], run(L.p2w_voxel_sample_table_prepared, *(b1, b2)[which], cells, wt, status)
        assert st == 0 and rst == 0
        m = int(r["ptr"][-1])
        assert torch.equal(o["ptr"], r["ptr"]) and torch.equal(o["grid"], r["grid"]) and torch.equal(o["inv"], r["inv"]), step
        for k in ("idx", "batch", "ckeys"):
            assert torch.equal(o[k][:m], r[k][:m]), (step, k)
        assert torch.equal(o["skeys"], r["skeys"]) and torch.equal(o["rsort"], o["inv"][o["order"].long()]), step
        T = int(r["grid"][4] * r["grid"][5] * r["grid"][6]) * (2 if which else 3)
        assert torch.equal(o["cs"][: T + 1], r["cs"][: T + 1]) and torch.equal(o["css"][: T + 1], r["css"][: T + 1]), step


@pytest.mark.parametrize("box", [0, 4])
@pytest.mark.parametrize("sizes,surface", [([5000], False), ([1500, 40, 2600], True), ([16384, 3000], False), ([300, 0, 9000], True)])
def test_indexed_grid_searches_equal_the_bisecting_ones(sizes, surface, box):
    """p2w_knn_grid_indexed / p2w_ball_query_grid_indexed with the table sampler's cell -> position tables (the engine's searches)
    give the results of p2w_knn_grid / p2w_ball_query_grid bit for bit: interpolation search (level 0 -> level 1, k = 2), SA-style
    search among the level itself (k = 32, queries = a coarser sample) and the ball query over the cell-sorted input points."""
    from pointstowood_amd._lib import SEARCH_Q_ROW_IN_W, SEARCH_X_INDEX_IN_W, lib, ptr, stream
    vox = [(synth.surface_voxel if surface else synth.uniform_voxel)(2.0, max(n, 1), 91 + i, True) for i, n in enumerate(sizes)]
    b = synth.collate(vox)
    keep = torch.cat([torch.ones(max(n, 1), dtype=torch.bool) if n > 0 else torch.zeros(1, dtype=torch.bool) for n in sizes])
    pos = b["pos"][keep].cuda()
    n, B, res = pos.shape[0], len(sizes), 0.04
    L = lib()
    xyzr = torch.zeros((n, 4), device="cuda")
    xyzr[:, :3] = pos
    csr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device="cuda")
    i32, i64 = dict(dtype=torch.int32, device="cuda"), dict(dtype=torch.int64, device="cuda")
    idx, ptr_out, batch_out, order = torch.empty(n, **i32), torch.empty(B + 1, **i32), torch.empty(n, **i32), torch.empty(n, **i32)
    skeys, ckeys, grid = torch.empty(n, **i64), torch.empty(n, **i64), torch.zeros(8, **i64)
    cells = B * (int(4.2 / res) + 3) ** 3      # surface voxels are centred on their mean: the batch box exceeds 2.3 m
    wt = torch.empty(int(L.p2w_voxel_sample_table_ws_bytes(n, cells)), dtype=torch.uint8, device="cuda")
    cs, css, status = torch.empty(cells + 1, **i32), torch.empty(cells + 1, **i32), torch.zeros(1, **i32)
    assert L.p2w_voxel_sample_table(ptr(xyzr), ptr(csr), B, n, res, ptr(idx), ptr(ptr_out), ptr(batch_out), ptr(order), ptr(skeys),
                                    ptr(ckeys), ptr(grid), None, None, ptr(cs), ptr(css), ptr(status), cells, ptr(wt), wt.numel(),
                                    stream()) == 0
    assert int(status) == 0
    m = int(ptr_out[B])
    rec = torch.empty((n, 4), device="cuda")
    assert L.p2w_index_records(ptr(xyzr), ptr(order), ptr(csr), B, n, ptr(rec), stream()) == 0
    coarse = xyzr[idx[:m].long()].contiguous()
    new = lambda rows, k: (torch.full((rows, k), -7, **i32), torch.full((rows,), -7, **i32))
    # k = 2, queries = all points in cell order, candidates = the sampled level
    (a_n, a_d), (b_n, b_d) = new(n, 2), new(n, 2)
    assert L.p2w_knn_grid(ptr(coarse), ptr(ckeys), ptr(ptr_out), ptr(grid), ptr(rec), None, ptr(csr), B, n, 2, ptr(a_n), ptr(a_d), None,
                          SEARCH_Q_ROW_IN_W | box, stream()) == 0
    assert L.p2w_knn_grid_indexed(ptr(coarse), ptr(ckeys), ptr(ptr_out), ptr(grid), ptr(cs), ptr(rec), None, ptr(csr), B, n, 2, ptr(b_n),
                                  ptr(b_d), None, SEARCH_Q_ROW_IN_W | box, stream()) == 0
    assert torch.equal(a_n, b_n) and torch.equal(a_d, b_d)
    # k = 32 among the sampled level itself, queries = every third of them
    q = torch.arange(0, m, 3, **i32)
    pq = torch.searchsorted(batch_out[:m][q.long()].long().contiguous(), torch.arange(B + 1, device="cuda")).int()
    (a_n, a_d), (b_n, b_d) = new(q.numel(), 32), new(q.numel(), 32)
    assert L.p2w_knn_grid(ptr(coarse), ptr(ckeys), ptr(ptr_out), ptr(grid), ptr(coarse), ptr(q), ptr(pq), B, q.numel(), 32, ptr(a_n),
                          ptr(a_d), None, box, stream()) == 0
    assert L.p2w_knn_grid_indexed(ptr(coarse), ptr(ckeys), ptr(ptr_out), ptr(grid), ptr(cs), ptr(coarse), ptr(q), ptr(pq), B, q.numel(),
                                  32, ptr(b_n), ptr(b_d), None, box, stream()) == 0
    assert torch.equal(a_n, b_n) and torch.equal(a_d, b_d)
    # ball query: candidates = the input points in cell-sorted order (carrying their own index), queries = the sampled level
    (a_n, a_d), (b_n, b_d) = new(m, 32), new(m, 32)
    assert L.p2w_ball_query_grid(ptr(rec), ptr(skeys), ptr(csr), ptr(grid), ptr(xyzr), ptr(idx), ptr(ptr_out), B, m, 0.08, 32, ptr(a_n),
                                 ptr(a_d), SEARCH_X_INDEX_IN_W | box, stream()) == 0
    assert L.p2w_ball_query_grid_indexed(ptr(rec), ptr(skeys), ptr(csr), ptr(grid), ptr(css), ptr(xyzr), ptr(idx), ptr(ptr_out), B, m, 0.08,
                                         32, ptr(b_n), ptr(b_d), SEARCH_X_INDEX_IN_W | box, stream()) == 0
    assert torch.equal(a_n, b_n) and torch.equal(a_d, b_d)
    if surface:
        assert int(a_d.max()) == 32


@pytest.mark.parametrize("sizes,surface", [([5000], False), ([1500, 40, 2600], True), ([16384, 3000], False)])
def test_voxel_sample_order_is_cell_sorted_permutation(sizes, surface):
    b = _batch(sizes, seed=21, surface=surface)
    s = _sorted_level(b, 0.04)
    order = s["order"].cpu().long()
    assert torch.equal(torch.sort(order).values, torch.arange(s["n"]))
    cell = O.voxel_grid(b["pos"], 0.04, b["batch"])
    assert bool((cell[order][1:] >= cell[order][:-1]).all())          # ascending (voxel, cell id)
    assert torch.equal(s["rec"][:, :3].cpu(), b["pos"][order])
    assert torch.equal(s["rec"][:, 3].contiguous().view(torch.int32).cpu().long(), order)


@pytest.mark.parametrize("sizes,cap,surface", [([4000], 32, False), ([3000], 8, True), ([1500, 40, 2600], 16, True),
                                               ([16384, 3000], 32, False)])
def test_ball_query_over_sorted_candidates_matches_plain(sizes, cap, surface):
    """P2W_SEARCH_X_INDEX_IN_W + tile boxes (the engine's SA1 search) == the plain search == the oracle."""
    from pointstowood_amd._lib import SEARCH_X_INDEX_IN_W, ptr, stream
    b = _batch(sizes, seed=23, surface=surface)
    s = _sorted_level(b, 0.04)
    L, m = s["L"], s["m"]
    out = []
    for x, box, flags in ((s["xyzr"], None, 0), (s["rec"], s["box"], SEARCH_X_INDEX_IN_W), (s["rec"], None, SEARCH_X_INDEX_IN_W)):
        nbr = torch.empty((m, cap), dtype=torch.int32, device="cuda")
        deg = torch.empty(m, dtype=torch.int32, device="cuda")
        assert L.p2w_ball_query(ptr(x), ptr(s["csr"]), ptr(s["xyzr"]), ptr(s["idx"]), ptr(s["ptr_out"]), s["B"], m, 0.08, cap,
                                ptr(nbr), ptr(deg), ptr(box), flags, stream()) == 0
        out.append((nbr.cpu(), deg.cpu()))
    for nbr, deg in out[1:]:
        assert torch.equal(nbr, out[0][0]) and torch.equal(deg, out[0][1])
    idx = s["idx"].cpu().long()
    ref = O.radius(b["pos"], b["pos"][idx], 0.08, b["batch"], b["batch"][idx], max_num_neighbors=cap)
    nbr, deg = out[1]
    mask = torch.arange(cap)[None, :] < deg[:, None]
    got = torch.stack([torch.arange(m)[:, None].expand(m, cap)[mask], nbr[mask].long()], 0)
    assert torch.equal(got, ref)
    if surface:
        assert int(deg.max()) == cap                                   # the cap is exercised


@pytest.mark.parametrize("sizes,k", [([5000], 2), ([1500, 40, 2600], 2), ([16384, 3000], 3)])
def test_knn_with_sorted_queries_writes_own_rows(sizes, k):
    """P2W_SEARCH_Q_ROW_IN_W (the engine's last interpolation search): level-0 queries visited in cell order."""
    from pointstowood_amd._lib import SEARCH_Q_ROW_IN_W, ptr, stream
    b = _batch(sizes, seed=29)
    s = _sorted_level(b, 0.04)
    L, n, B = s["L"], s["n"], s["B"]
    coarse = s["xyzr"][s["idx"].long()].contiguous()                   # level-1 records (no scale round trip here)
    cbox = torch.empty((L.p2w_tile_bbox_count(B, s["m"]), 6), dtype=torch.float32, device="cuda")
    assert L.p2w_tile_bbox(ptr(coarse), ptr(s["ptr_out"]), B, s["m"], ptr(cbox), stream()) == 0
    res = []
    for q, flags in ((s["xyzr"], 0), (s["rec"], SEARCH_Q_ROW_IN_W)):
        nbr = torch.full((n, k), -7, dtype=torch.int32, device="cuda")
        deg = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        assert L.p2w_knn(ptr(coarse), ptr(s["ptr_out"]), ptr(q), None, ptr(s["csr"]), B, n, k, ptr(nbr), ptr(deg), ptr(cbox),
                         flags, stream()) == 0
        res.append((nbr.cpu(), deg.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    idx = s["idx"].cpu().long()
    ref = O.knn(b["pos"][idx], b["pos"], k, b["batch"][idx], b["batch"])
    nbr, deg = res[1]
    mask = torch.arange(k)[None, :] < deg[:, None]
    got = torch.stack([torch.arange(n)[:, None].expand(n, k)[mask], nbr[mask].long()], 0)
    assert torch.equal(got, ref)


def _grid_fields(grid):
    raw = grid.cpu().numpy().view(np.uint8)
    f = raw[:32].view(np.float32)
    return dict(lo=f[0:3], res=float(f[3]), hi=f[4:7], b_lo=int(raw[28:32].view(np.int32)[0]), dims=raw[32:56].view(np.int64))


@pytest.mark.parametrize("sizes,surface", [([5000], False), ([1500, 40, 2600], True), ([0, 300, 0, 16384], False)])
def test_voxel_sample_search_index(sizes, surface):
    """sorted keys / representative keys / grid geometry written by p2w_voxel_sample == PyG voxel_grid's cell ids."""
    vox = [(synth.surface_voxel if surface else synth.uniform_voxel)(2.0, n, 31 + i, True) for i, n in enumerate(sizes) if n > 0]
    b = synth.collate(vox)
    if 0 in sizes:   # empty voxels in between: batch ids of the non-empty ones
        ids = torch.tensor([i for i, n in enumerate(sizes) if n > 0])
        b["batch"] = ids[b["batch"]]
    s = _sorted_level(b, 0.04)
    cell = O.voxel_grid(b["pos"], 0.04, b["batch"])
    assert torch.equal(s["skeys"].cpu(), torch.sort(cell).values)
    assert torch.equal(s["ckeys"].cpu(), torch.unique(cell))
    g = _grid_fields(s["grid"])
    pos = b["pos"].numpy()
    assert np.array_equal(g["lo"], pos.min(0)) and np.array_equal(g["hi"], pos.max(0)) and g["res"] == np.float32(0.04)
    assert g["b_lo"] == int(b["batch"].min())
    assert np.array_equal(g["dims"], ((pos.max(0) - pos.min(0)) / np.float32(0.04)).astype(np.int64) + 1)


def _level1(s):
    """level-1 records (plain gather), their CSR, keys and boxes for the searches below."""
    from pointstowood_amd._lib import ptr, stream
    L, B, m = s["L"], s["B"], s["m"]
    coarse = s["xyzr"][s["idx"].long()].contiguous()
    cbox = torch.empty((L.p2w_tile_bbox_count(B, max(m, 1)), 6), dtype=torch.float32, device="cuda")
    assert L.p2w_tile_bbox(ptr(coarse), ptr(s["ptr_out"]), B, m, ptr(cbox), stream()) == 0
    return coarse, cbox


@pytest.mark.parametrize("box", [0, 4])   # 4 = P2W_SEARCH_BOX: region bounded in x too (one run per grid row)
@pytest.mark.parametrize("sizes,k,surface", [([5000], 2, False), ([1500, 40, 2600], 2, True), ([16384, 3000], 3, False),
                                             ([9000], 32, True), ([700, 5, 16384], 32, False), ([60], 64, False)])
def test_knn_grid_matches_brute_force_other_level_queries(sizes, k, surface, box):
    """p2w_knn_grid, queries = level 0 in cell order (row-in-w), candidates = level 1: the interpolation searches."""
    from pointstowood_amd._lib import SEARCH_Q_ROW_IN_W, ptr, stream
    b = _batch(sizes, seed=37, surface=surface)
    s = _sorted_level(b, 0.04)
    L, n, B = s["L"], s["n"], s["B"]
    coarse, cbox = _level1(s)
    out = []
    for grid in (False, True):
        nbr = torch.full((n, k), -7, dtype=torch.int32, device="cuda")
        deg = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        if grid:
            st = L.p2w_knn_grid(ptr(coarse), ptr(s["ckeys"]), ptr(s["ptr_out"]), ptr(s["grid"]), ptr(s["rec"]), None,
                                ptr(s["csr"]), B, n, k, ptr(nbr), ptr(deg), None, SEARCH_Q_ROW_IN_W | box, stream())
        else:
            st = L.p2w_knn(ptr(coarse), ptr(s["ptr_out"]), ptr(s["xyzr"]), None, ptr(s["csr"]), B, n, k, ptr(nbr), ptr(deg),
                           ptr(cbox), 0, stream())
        assert st == 0
        out.append((nbr.cpu(), deg.cpu()))
    assert torch.equal(out[0][1], out[1][1])
    assert torch.equal(out[0][0], out[1][0])


@pytest.mark.parametrize("sizes,k,res2,surface", [([5000], 32, 0.08, False), ([3000, 33, 9000], 32, 0.08, True),
                                                  ([16384], 16, 0.16, False), ([16384, 16384], 32, 0.08, True)])
@pytest.mark.parametrize("box", [0, 4])
def test_knn_grid_matches_brute_force_subset_queries(sizes, k, res2, surface, box):
    """p2w_knn_grid, queries = a coarser sample of the candidates themselves (qidx): the SA2 / SA3 searches."""
    from pointstowood_amd._lib import ptr, stream
    b = _batch(sizes, seed=41, surface=surface)
    s = _sorted_level(b, 0.04)
    L, B = s["L"], s["B"]
    coarse, cbox = _level1(s)
    m1 = s["m"]
    # second sampling over level 1 gives the query subset (indices into level 1) in its own cell order
    i32 = dict(dtype=torch.int32, device="cuda")
    idx2, ptr2, batch2 = torch.empty(m1, **i32), torch.empty(B + 1, **i32), torch.empty(m1, **i32)
    ws = torch.empty(int(L.p2w_voxel_sample_ws_bytes(m1)), dtype=torch.uint8, device="cuda")
    assert L.p2w_voxel_sample(ptr(coarse), ptr(s["ptr_out"]), B, m1, res2, ptr(idx2), ptr(ptr2), ptr(batch2), None, None, None,
                              None, None, None, ptr(ws), ws.numel(), stream()) == 0
    m2 = int(ptr2[B])
    out = []
    for grid in (False, True):
        nbr = torch.full((m2, k), -7, dtype=torch.int32, device="cuda")
        deg = torch.full((m2,), -7, dtype=torch.int32, device="cuda")
        if grid:
            st = L.p2w_knn_grid(ptr(coarse), ptr(s["ckeys"]), ptr(s["ptr_out"]), ptr(s["grid"]), ptr(coarse), ptr(idx2),
                                ptr(ptr2), B, m2, k, ptr(nbr), ptr(deg), None, box, stream())
        else:
            st = L.p2w_knn(ptr(coarse), ptr(s["ptr_out"]), ptr(coarse), ptr(idx2), ptr(ptr2), B, m2, k, ptr(nbr), ptr(deg),
                           ptr(cbox), 0, stream())
        assert st == 0
        out.append((nbr.cpu(), deg.cpu()))
    assert torch.equal(out[0][1], out[1][1])
    assert torch.equal(out[0][0], out[1][0])


@pytest.mark.parametrize("sizes,k,surface,dup", [([5000], 32, False, False), ([3000, 33, 9000], 32, True, False), ([16384], 8, False, False),
                                                 ([2000, 16384], 64, True, False), ([4000], 32, False, True), ([70], 64, False, True)])
@pytest.mark.parametrize("box", [0, 4])
def test_knn_grid_on_duplicates_and_short_voxels_equals_brute_force(sizes, k, surface, dup, box):
    """The grid kNN's threshold ladder + sorted insertions against the brute-force kernel, bit for bit - k = 8 .. 64, voxels with
    fewer points than k, and clouds of exact duplicates (every distance a four-fold (d2, index) tie: only the index order decides)."""
    from pointstowood_amd._lib import ptr, stream
    b = _batch(sizes, seed=43, surface=surface)
    if dup:   # a quarter of the points, each four times
        pos = b["pos"].clone()
        o = 0
        for n in sizes:
            q = max(1, n // 4)
            pos[o:o + n] = pos[o:o + q].clone().repeat(-(-n // q), 1)[:n]
            o += n
        b = dict(b, pos=pos)
    s = _sorted_level(b, 0.04)
    L, B = s["L"], s["B"]
    coarse, _ = _level1(s)
    m1 = s["m"]
    out = []
    for grid in (True, False):
        nbr = torch.full((m1, k), -7, dtype=torch.int32, device="cuda")
        deg = torch.full((m1,), -7, dtype=torch.int32, device="cuda")
        if grid:
            st = L.p2w_knn_grid(ptr(coarse), ptr(s["ckeys"]), ptr(s["ptr_out"]), ptr(s["grid"]), ptr(coarse), None, ptr(s["ptr_out"]),
                                B, m1, k, ptr(nbr), ptr(deg), None, box, stream())
        else:
            st = L.p2w_knn(ptr(coarse), ptr(s["ptr_out"]), ptr(coarse), None, ptr(s["ptr_out"]), B, m1, k, ptr(nbr), ptr(deg), None, 0, stream())
        assert st == 0
        out.append((nbr.cpu(), deg.cpu()))
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][0], out[1][0])
    assert int(out[0][1].min()) >= 1 and bool((out[0][0][:, 0] == torch.arange(m1)).all() or dup)   # self first (unique points)


@pytest.mark.parametrize("sizes,cap,surface", [([4000], 32, False), ([3000], 8, True), ([1500, 40, 2600], 16, True),
                                               ([16384, 3000], 32, False)])
@pytest.mark.parametrize("box", [0, 4])
def test_ball_query_grid_matches_brute_force(sizes, cap, surface, box):
    from pointstowood_amd._lib import SEARCH_X_INDEX_IN_W, ptr, stream
    b = _batch(sizes, seed=23, surface=surface)
    s = _sorted_level(b, 0.04)
    L, m = s["L"], s["m"]
    out = []
    for grid in (False, True):
        nbr = torch.full((m, cap), -7, dtype=torch.int32, device="cuda")
        deg = torch.full((m,), -7, dtype=torch.int32, device="cuda")
        if grid:
            st = L.p2w_ball_query_grid(ptr(s["rec"]), ptr(s["skeys"]), ptr(s["csr"]), ptr(s["grid"]), ptr(s["xyzr"]),
                                       ptr(s["idx"]), ptr(s["ptr_out"]), s["B"], m, 0.08, cap, ptr(nbr), ptr(deg),
                                       SEARCH_X_INDEX_IN_W | box, stream())
        else:
            st = L.p2w_ball_query(ptr(s["xyzr"]), ptr(s["csr"]), ptr(s["xyzr"]), ptr(s["idx"]), ptr(s["ptr_out"]), s["B"], m,
                                  0.08, cap, ptr(nbr), ptr(deg), None, 0, stream())
        assert st == 0
        out.append((nbr.cpu(), deg.cpu()))
    assert torch.equal(out[0][1], out[1][1])
    assert torch.equal(out[0][0], out[1][0])


@pytest.mark.parametrize("box", [0, 4])
@pytest.mark.parametrize("res", [0.04, 0.01])
def test_knn_grid_far_apart_clusters_and_duplicates(res, box):
    """Queries whose neighbours are many cells away (region growth; at res 0.01 the region outgrows the 128-layer run
    table and the kernel falls back to the whole voxel) and coincident points (ties)."""
    from pointstowood_amd._lib import SEARCH_Q_ROW_IN_W, ptr, stream
    g = torch.Generator().manual_seed(5)
    a = torch.rand(300, 3, generator=g) * 0.2                 # dense clump in one corner
    far = torch.rand(40, 3, generator=g) * 0.05 + 1.9          # a few points in the opposite corner
    dup = a[:50].clone()                                       # exact duplicates
    pos = torch.cat([a, far, dup], 0)
    pos = pos[torch.randperm(pos.shape[0], generator=g)]
    b = dict(pos=pos, batch=torch.zeros(pos.shape[0], dtype=torch.long))
    s = _sorted_level(b, res)
    L, n, B = s["L"], s["n"], s["B"]
    coarse, cbox = _level1(s)
    for k in (2, 32, 64):
        out = []
        for grid in (False, True):
            nbr = torch.full((n, k), -7, dtype=torch.int32, device="cuda")
            deg = torch.full((n,), -7, dtype=torch.int32, device="cuda")
            if grid:
                st = L.p2w_knn_grid(ptr(coarse), ptr(s["ckeys"]), ptr(s["ptr_out"]), ptr(s["grid"]), ptr(s["rec"]), None,
                                    ptr(s["csr"]), B, n, k, ptr(nbr), ptr(deg), None, SEARCH_Q_ROW_IN_W | box, stream())
            else:
                st = L.p2w_knn(ptr(coarse), ptr(s["ptr_out"]), ptr(s["xyzr"]), None, ptr(s["csr"]), B, n, k, ptr(nbr),
                               ptr(deg), ptr(cbox), 0, stream())
            assert st == 0
            out.append((nbr.cpu(), deg.cpu()))
        assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][0], out[1][0])


def test_searches_with_nan_queries_report_nothing_for_them():
    """A query with NaN coordinates admits no candidate: deg 0 / all -1 for it, every other query unaffected."""
    from pointstowood_amd._lib import ptr, stream
    b = _batch([4000], seed=43)
    s = _sorted_level(b, 0.04)
    L, B = s["L"], s["B"]
    coarse, cbox = _level1(s)
    q = s["xyzr"][:200].clone()
    bad = torch.tensor([3, 77, 199], device="cuda")
    q[bad, 1] = float("nan")
    pq = torch.tensor([0, 200], dtype=torch.int32, device="cuda")
    clean = torch.ones(200, dtype=torch.bool)
    clean[bad.cpu()] = False
    ref = None
    for k in (2, 32):
        for grid in (False, True):
            nbr = torch.full((200, k), -7, dtype=torch.int32, device="cuda")
            deg = torch.full((200,), -7, dtype=torch.int32, device="cuda")
            if grid:
                st = L.p2w_knn_grid(ptr(coarse), ptr(s["ckeys"]), ptr(s["ptr_out"]), ptr(s["grid"]), ptr(q), None, ptr(pq), B,
                                    200, k, ptr(nbr), ptr(deg), None, 0, stream())
            else:
                st = L.p2w_knn(ptr(coarse), ptr(s["ptr_out"]), ptr(q), None, ptr(pq), B, 200, k, ptr(nbr), ptr(deg), ptr(cbox),
                               0, stream())
            assert st == 0
            nbr, deg = nbr.cpu(), deg.cpu()
            assert bool((deg[~clean] == 0).all()) and bool((nbr[~clean] == -1).all())
            assert bool((deg[clean] == k).all())
            if grid:
                assert torch.equal(nbr[clean], ref)
            else:
                ref = nbr[clean]


def test_grid_searches_randomised_stress():
    """tools/stress_search.py: 150 random geometries (clusters, planes, duplicates, lattices, far offsets, tiny and huge
    cells, BOX on/off, k in 1..64): grid kNN / ball query must equal the whole-voxel kernels bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_search.py"), "150", "7"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("sizes,surface", [([5000], False), ([1500, 40, 2600], True), ([16384, 3000], False), ([3], False)])
def test_knn_grid_with_sampler_hints_and_with_bogus_hints(sizes, surface):
    """k = 2 searches seeded by p2w_knn_hint2 (the engine's interpolation searches) == unseeded == brute force; hints that
    are far too small must be detected and cost only a rescan."""
    from pointstowood_amd._lib import SEARCH_Q_ROW_IN_W, lib, ptr, stream
    b = _batch(sizes, seed=51, surface=surface)
    L = lib()
    pos, batch = b["pos"].cuda(), b["batch"].cuda()
    n, B = pos.shape[0], int(batch.max()) + 1
    xyzr = torch.zeros((n, 4), dtype=torch.float32, device="cuda"); xyzr[:, :3] = pos
    csr = torch.zeros(B + 1, dtype=torch.int32, device="cuda")
    csr[1:] = torch.cumsum(torch.bincount(batch, minlength=B), 0).int()
    i32 = dict(dtype=torch.int32, device="cuda")
    idx, ptr_out, bo, order, inv, rks = (torch.empty(n, **i32) for _ in range(6))
    ptr_out = torch.empty(B + 1, **i32)
    ckeys = torch.empty(n, dtype=torch.int64, device="cuda"); grid = torch.zeros(8, dtype=torch.int64, device="cuda")
    ws = torch.empty(int(L.p2w_voxel_sample_ws_bytes(n)), dtype=torch.uint8, device="cuda")
    assert L.p2w_voxel_sample(ptr(xyzr), ptr(csr), B, n, 0.04, ptr(idx), ptr(ptr_out), ptr(bo), ptr(order), None, ptr(ckeys),
                              ptr(grid), ptr(inv), ptr(rks), ptr(ws), ws.numel(), stream()) == 0
    m = int(ptr_out[B])
    # inv / rank_sorted against the oracle's consecutive_cluster
    inv_ref, _ = O.consecutive_cluster(O.voxel_grid(b["pos"], 0.04, b["batch"]))
    assert torch.equal(inv.cpu().long(), inv_ref) and torch.equal(rks.cpu().long(), inv_ref[order.cpu().long()])
    coarse = xyzr[idx[:m].long()].contiguous()
    rec = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    assert L.p2w_index_records(ptr(xyzr), ptr(order), ptr(csr), B, n, ptr(rec), stream()) == 0
    hint = torch.empty(n, dtype=torch.float32, device="cuda")
    assert L.p2w_knn_hint2(ptr(rec), ptr(rks), ptr(csr), B, n, ptr(coarse), ptr(hint), stream()) == 0
    res = []
    for h in (None, hint, torch.full_like(hint, 1e-12), torch.where(torch.arange(n, device="cuda") % 3 == 0, hint, hint * 0 + float("inf"))):
        nbr = torch.full((n, 2), -7, **i32); deg = torch.full((n,), -7, **i32)
        assert L.p2w_knn_grid(ptr(coarse), ptr(ckeys), ptr(ptr_out), ptr(grid), ptr(rec), None, ptr(csr), B, n, 2, ptr(nbr),
                              ptr(deg), ptr(h), SEARCH_Q_ROW_IN_W, stream()) == 0
        res.append((nbr.cpu(), deg.cpu()))
    for r in res[1:]:
        assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1])
    # the hint really is an upper bound of the 2nd-nearest distance wherever it is finite
    nb = res[0][0].long()
    ok = (res[0][1] == 2)
    q = b["pos"]
    d2nd = ((q[ok] - coarse.cpu()[nb[ok][:, 1], :3]) ** 2).sum(1)
    hq = torch.empty(n); hq[order.cpu().long()] = hint.cpu()
    assert bool((d2nd <= hq[ok] * (1 + 1e-5) + 1e-12).all())


# ---- the 8th operator (MessagePassing.propagate, aggr = max) and the fused PointNetConv layer ----------------------------
def _sa_level_inputs(sizes, res, k, seed, surface=False, F_in=8):
    """One SA level's inputs as SAModule.forward builds them (model.py:109-123): sources, sampled targets, edges."""
    b = _batch(sizes, seed=seed, surface=surface)
    pos4 = torch.cat([b["pos"], b["reflectance"][:, None]], 1)
    idx = O.consecutive_cluster(O.voxel_grid(b["pos"], res, b["batch"]))[1]
    if res == 0.04:
        row, col = O.radius(b["pos"], b["pos"][idx], res * 2, b["batch"], b["batch"][idx], max_num_neighbors=k)
    else:
        row, col = O.knn(b["pos"], b["pos"][idx], k, b["batch"], b["batch"][idx])
    pos4 = pos4.clone()
    pos4[:, :3] = pos4[:, :3] / b["sf"][b["batch"]][:, None]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(pos4.shape[0], F_in, generator=g)
    return x, pos4, idx, torch.stack([col, row], 0)


def _local_nn(F_in, C1, C2, seed):
    """MLP([F_in + 4, C1, C2]) as model.py:198-202 builds it, with non-trivial BN statistics (negative scales included)."""
    from torch.nn import BatchNorm1d as BN, Linear as Lin, ReLU, Sequential as Seq
    torch.manual_seed(seed)
    nn = Seq(Seq(Lin(F_in + 4, C1), ReLU()), Seq(Lin(C1, C2), ReLU(), BN(C2)))
    bn = nn[1][2]
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C2)); bn.bias.copy_(torch.randn(C2) * 0.3)
        bn.running_mean.copy_(torch.randn(C2) * 0.2); bn.running_var.copy_(torch.rand(C2) + 0.5)
    return nn.eval()


def _message_reference(nn, x, pos_src, pos_dst, edge_index):
    """PointNetConv.message + max aggregation (pointnet.py:111-132) in plain PyTorch on the CPU."""
    j, i = edge_index
    rel = pos_src[j, :3] - pos_dst[i, :3]
    nrm = torch.norm(rel, dim=1, keepdim=True)
    dmax = O.scatter_max(nrm, i, dim=0, dim_size=pos_dst.shape[0])[0]
    msg = torch.zeros(j.numel(), 4)
    msg[:, :3] = rel / (dmax[i] + 1e-8)
    msg[:, 3] = pos_src[j, 3]
    with torch.no_grad():
        out = nn(torch.cat([x[j], msg], 1))
    return O.segment_max_rows(out, i, pos_dst.shape[0])


@pytest.mark.parametrize("sizes,res,k,surface", [([2000], 0.08, 32, False), ([1500, 300], 0.04, 32, True), ([900, 40], 0.16, 16, False)])
def test_pointnet_conv_operator_matches_the_message_passing_definition(H, sizes, res, k, surface):
    """ops.PointNetConv (fused p2w_gemm + p2w_sa_conv) == gather -> local_nn -> max, the layer of pointnet.py:86-132,
    called the way model.py:123 calls it: conv(x, (pos, pos[idx]), edge_index)."""
    F_in, C1, C2 = 8, 16, 32
    x, pos4, idx, ei = _sa_level_inputs(sizes, res, k, seed=11, surface=surface, F_in=F_in)
    nn = _local_nn(F_in, C1, C2, seed=5)
    ref = _message_reference(nn, x, pos4, pos4[idx], ei)
    import copy
    conv = H.PointNetConv(local_nn=copy.deepcopy(nn), global_nn=None, add_self_loops=False, radius=res).cuda().eval()
    assert [n for n, _ in conv.state_dict().items()][:2] == ["local_nn.0.0.weight", "local_nn.0.0.bias"]   # the reference's key names
    got = conv(x.cuda(), (pos4.cuda(), pos4[idx].cuda()), ei.cuda())
    assert got.shape == ref.shape
    assert (got.cpu() - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_message_passing_propagate_runs_the_reference_style_subclass(H):
    """A PointNetConv written like the reference's (pointnet.py:19-132: subclass of MessagePassing with message(x_j, pos_i,
    pos_j, edge_index_i), aggr = max) over ops.MessagePassing + ops.scatter_max: the unfused form of the same layer."""
    F_in, C1, C2 = 8, 16, 32
    x, pos4, idx, ei = _sa_level_inputs([1200, 200], 0.08, 32, seed=13, F_in=F_in)
    nn = _local_nn(F_in, C1, C2, seed=6)

    class RefStyleConv(H.MessagePassing):
        def __init__(self, local_nn):
            super().__init__(aggr="max")
            self.local_nn = local_nn

        def forward(self, x, pos, edge_index):
            return self.propagate(edge_index, x=(x, None), pos=pos)

        def message(self, x_j, pos_i, pos_j, edge_index_i):
            msg = torch.zeros((pos_j.size(0), pos_j.size(1)), device=pos_j.device)
            relative_pos = pos_j[:, :3] - pos_i[:, :3]
            max_distances, _ = H.scatter_max(torch.norm(relative_pos, dim=1, keepdim=True), edge_index_i, dim=0)
            msg[:, :3] = relative_pos / (max_distances[edge_index_i] + 1e-8)
            msg[:, 3] = pos_j[:, 3]
            return self.local_nn(torch.cat([x_j, msg], dim=1))

    ref = _message_reference(nn, x, pos4, pos4[idx], ei)
    import copy
    conv = RefStyleConv(copy.deepcopy(nn)).cuda().eval()
    with torch.no_grad():
        got = conv(x.cuda(), (pos4.cuda(), pos4[idx].cuda()), ei.cuda())
    assert (got.cpu() - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("k,F", [(3, 24), (5, 10), (1, 7)])
def test_knn_interpolate_general_k_and_width(H, k, F):
    """PyG's default k = 3 (and other k, and widths that are not multiples of 4) against the oracle."""
    b = _batch([2500, 60, 900], seed=21)
    idx = O.consecutive_cluster(O.voxel_grid(b["pos"], 0.16, b["batch"]))[1]
    feat = torch.randn(idx.numel(), F, generator=torch.Generator().manual_seed(4))
    ref = O.knn_interpolate(feat, b["pos"][idx], b["pos"], b["batch"][idx], b["batch"], k=k)
    got = H.knn_interpolate(feat.cuda(), b["pos"][idx].cuda(), b["pos"].cuda(), b["batch"][idx].cuda(), b["batch"].cuda(), k=k)
    assert got.shape == ref.shape
    assert (got.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()
