#!/usr/bin/env python3
"""p2w_gemm_h2 (whole tiles only) against p2w_gemm_h2_sk (stream-K tail: library's choice, and forced) on the shapes of the
forward whose tiles fill their last chip round badly, and on small-batch shapes: time per launch (median of 30, HIP events around
10 back-to-back launches) and max |difference| against fp64.   python tools/gemm_sk_ab.py [prec]"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_ops import _from_h, _pack_h, _to_h  # noqa: E402
from pointstowood_amd._lib import GEMM_NO_STREAMK, GEMM_STREAMK, Epilogue, check, lib, ptr, stream  # noqa: E402

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 0
SHAPES = [  # (M, K, N): level 3 unchunked, its remainder, FP4 / sa4, small batches
    (17506, 512, 2048), (17506, 2048, 2048), (17506, 2048, 512), (1122, 2048, 2048), (1122, 512, 2048), (1122, 2048, 512),
    (17506, 516, 512), (17506, 1024, 768), (17506, 768, 512), (123046, 512, 512), (81683, 768, 640),
    (10840, 128, 512), (10840, 512, 512), (7000, 1024, 1024), (2000, 2048, 2048), (2000, 2048, 512), (4096, 1024, 256), (300, 2048, 2048),
]
L = lib()
ws = torch.empty(int(L.p2w_gemm_h2_sk_ws_bytes()), dtype=torch.uint8, device="cuda")
print(f"{'M':>7s} {'K':>5s} {'N':>5s} | {'plain us':>9s} {'auto us':>9s} {'forced us':>9s} | {'TF plain':>8s} {'TF auto':>8s} | err plain / auto / forced (rel. to max |v|)")
for (M, K, N) in SHAPES:
    g = torch.Generator().manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    dW, wscale, Kp = _pack_h(W, prec)
    ka = 32 if prec == 0 else 64
    ldh_a = (K + 4 + ka - 1) // ka * ka
    Ah = _to_h(A, prec, ldh_a)
    db = bias.cuda()
    ep = Epilogue(ptr(db), None, None, None, None, None, 0, 1, 0, 0, 0)
    ldh_o = (N + ka - 1) // ka * ka
    planes = 2 if prec == 0 else 1
    v = torch.relu(A.double().cuda() @ W.double().cuda().t() + bias.double().cuda())
    scale = float(v.abs().max())
    row = []
    for flags in (GEMM_NO_STREAMK, 0, GEMM_STREAMK):
        out = torch.full((M, N), float("nan"), device="cuda")
        outh = torch.full((M, planes * ldh_o), float("nan"), dtype=Ah.dtype, device="cuda")

        def run():
            check(L.p2w_gemm_h2_sk(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(out), N, ptr(outh), ldh_o,
                                   ptr(ws), ws.numel(), flags, stream()))
        run()
        torch.cuda.synchronize()
        err = float((out.double() - v).abs().max()) / scale
        errh = float((_from_h(outh, prec, ldh_o)[:, :N].cuda() - v).abs().max()) / scale
        ts = []
        for rep in range(30):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                run()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 100)
        row.append((statistics.median(ts), max(err, errh)))
    tf = lambda t: 2.0 * M * K * N / (t * 1e-6) / 1e12
    print(f"{M:7d} {K:5d} {N:5d} | {row[0][0]:9.1f} {row[1][0]:9.1f} {row[2][0]:9.1f} | {tf(row[0][0]):8.0f} {tf(row[1][0]):8.0f} | "
          f"{row[0][1]:.1e} / {row[1][1]:.1e} / {row[2][1]:.1e}", flush=True)
