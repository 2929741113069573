#!/usr/bin/env python3
"""Randomised shapes / epilogues / tile flags for p2w_gemm_h2 against fp64 (MI355X).  usage: tools/stress_gemm.py [cases] [seed]"""
import ctypes as C
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_ops import H_TOL, _from_h, _pack_h, _to_h  # noqa: E402
from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream  # noqa: E402

cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
worst = {0: 0.0, 1: 0.0, 2: 0.0}
for case in range(cases):
    prec = rng.choice([0, 0, 1, 2])
    M = rng.choice([1, 7, 63, 64, 255, 256, 257, 300, 511, 777, 1024, 2100, 4097, 9000])
    N = rng.choice([1, 2, 3, 30, 64, 65, 128, 130, 192, 256, 258, 384, 512, 640, 768])
    K = rng.choice([4, 32, 36, 64, 100, 128, 200, 256, 512, 516, 1024])
    flags = rng.choice([0, 0, 0, 1, 2, 4, 8, 16, 1 | 16, 2 | 8, 1 << 24, (1 << 24) | 4, (1 << 24) | 16, 1 << 25])   # (1 << 24: the 64 x 128 tile; 1 << 25: never)
    use = {k: rng.random() < 0.6 for k in ("bias", "s0", "s1", "res", "f32", "h")}
    if not (use["f32"] or use["h"]):
        use["f32"] = True
    relu = [rng.randint(0, 1) for _ in range(4)]
    g = torch.Generator().manual_seed(case * 7919 + seed)
    Kc = (K + 3) // 4 * 4
    A = torch.randn(M, Kc, generator=g)
    A[:, K:] = 0
    W = torch.randn(N, K, generator=g) / K ** 0.5
    vec = lambda: torch.randn(N, generator=g)
    bias, s0, t0, s1, t1 = vec(), vec(), vec(), vec(), vec()
    R = torch.randn(M, N, generator=g)
    dW, wscale, Kp = _pack_h(W, prec)
    ka = 32 if prec == 0 else 64
    ldh_a = (Kc + 4 + ka - 1) // ka * ka
    Ah = _to_h(A, prec, ldh_a)
    d = lambda t: t.cuda().contiguous()
    db, ds0, dt0, ds1, dt1, dR = map(d, (bias, s0, t0, s1, t1, R))
    ep = Epilogue(ptr(db) if use["bias"] else None, ptr(ds0) if use["s0"] else None, ptr(dt0) if use["s0"] else None,
                  ptr(ds1) if use["s1"] else None, ptr(dt1) if use["s1"] else None, ptr(dR) if use["res"] else None, N, *relu)
    out = torch.full((M, N), float("nan"), device="cuda") if use["f32"] else None
    ldh_o = (N + ka - 1) // ka * ka
    planes = 2 if prec == 0 else 1
    outh = torch.full((M, planes * ldh_o), float("nan"), dtype=Ah.dtype, device="cuda") if use["h"] else None
    check(lib().p2w_gemm_h2(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(out), N, ptr(outh), ldh_o, flags, stream()))
    v = A[:, :K].double() @ W.double().t()
    if use["bias"]:
        v = v + bias.double()
    if relu[0]:
        v = torch.relu(v)
    if use["s0"]:
        v = v * s0.double() + t0.double()
    if relu[1]:
        v = torch.relu(v)
    if use["s1"]:
        v = v * s1.double() + t1.double()
    if relu[2]:
        v = torch.relu(v)
    if use["res"]:
        v = v + R.double()
    if relu[3]:
        v = torch.relu(v)
    scale = max(1.0, v.abs().max().item())
    tag = f"case {case}: prec {prec} M {M} N {N} K {K} flags {flags} {use} relu {relu}"
    if out is not None:
        err = (out.cpu().double() - v).abs().max().item()
        assert err <= H_TOL[prec] * scale, (tag, err)
        worst[prec] = max(worst[prec], err / scale)
    if outh is not None:
        hv = _from_h(outh, prec, ldh_o)
        tol = (H_TOL[prec] + (2e-6 if prec == 0 else 1e-3 if prec == 1 else 8e-3)) * scale
        assert (hv[:, :N] - v).abs().max().item() <= tol, (tag, (hv[:, :N] - v).abs().max().item())
        assert float(hv[:, N:ldh_o].abs().max() if ldh_o > N else 0.0) == 0.0, tag
print(f"{cases} cases ok; worst relative error f16x3 {worst[0]:.2e}, fp16 {worst[1]:.2e}, bf16 {worst[2]:.2e}")
