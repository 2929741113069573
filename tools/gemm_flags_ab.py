#!/usr/bin/env python3
"""Interleaved in-process A/B of p2w_gemm_h2 flag sets on the network's GEMM shapes (MI355X).

    python tools/gemm_flags_ab.py 0 8 16          # library tile order vs row-tile / column-slice order per XCD
    PREC=1 python tools/gemm_flags_ab.py 0 1 2    # fp16: library choice vs forced 128 / 256 tiles

Outputs of every flag set are compared with the first one's (the epilogue variants must agree bit for bit).
"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import _lib  # noqa: E402
from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream  # noqa: E402

flagsets = [int(v) for v in sys.argv[1:]] or [0, 32]
PREC = int(os.environ.get("PREC", "0"))
planes, ka = (2, 32) if PREC == 0 else (1, 64)
hdt = torch.bfloat16 if PREC == 2 else torch.float16
dev = torch.device("cuda")
# (M, K, N, out): the bench forward's layers (level sizes of BASELINE configs[1]); out: h = H tensor, f = fp32
shapes = [(131072, 32, 64, "f"), (123046, 128, 192, "f"), (81683, 256, 384, "f"),                  # hoists
          (123046, 128, 512, "h"), (123046, 512, 512, "h"), (123046, 512, 128, "f"),               # residual block, level 1
          (81683, 256, 1024, "h"), (81683, 1024, 1024, "h"), (81683, 1024, 256, "f"),              # level 2
          (17506, 512, 2048, "h"), (17506, 2048, 2048, "h"), (17506, 2048, 512, "f"),              # level 3 (whole)
          (16384, 2048, 2048, "h"), (1122, 2048, 2048, "h"), (16384, 2048, 512, "f"),              # level 3 (whole rounds + rest)
          (17506, 516, 512, "h"), (17506, 512, 512, "f"), (17506, 1024, 768, "h"), (17506, 768, 512, "f"),   # sa4, fp4
          (81683, 768, 640, "h"), (81683, 640, 512, "f"), (123046, 640, 512, "h"),                 # fp3, fp2
          (65536, 544, 512, "h"), (65536, 512, 512, "h")]                                          # fp1 / head
if os.environ.get("SHAPES"):
    shapes = [shapes[int(i)] for i in os.environ["SHAPES"].split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
for M, K, N, kind in shapes:
    Np, Kp = _lib.packed_dims(N, K, PREC)
    lda = Kp + int(os.environ.get("LDA_EXTRA", "0"))        # row pitch experiment: ldh_a may exceed K_pad
    A = torch.zeros(M, planes * lda, device=dev, dtype=hdt)
    A[:, : planes * K] = (torch.randn(M, planes * K, device=dev, generator=g) * 0.5).to(hdt)
    W = torch.zeros(Np, planes * Kp, device=dev, dtype=hdt)     # H rows: timing only, any finite content will do
    W[:N, : planes * K] = (torch.randn(N, planes * K, device=dev, generator=g) * 0.5).to(hdt)
    bias, sc, sh = (torch.randn(N, device=dev, generator=g) for _ in range(3))
    ldh_o = (N + ka - 1) // ka * ka
    ep = Epilogue(ptr(bias), ptr(sc), ptr(sh), None, None, None, 0, 1, 1, 0, 0)    # relu0, sc0, relu1 (class 263 / 135)
    if kind == "f":
        ep = Epilogue(ptr(bias), None, None, None, None, None, 0, 1, 0, 0, 0)      # bias + relu (class 129)
    outs = {}
    for fl in flagsets:
        outs[fl] = (torch.zeros(M, planes * ldh_o, dtype=hdt, device=dev) if kind == "h" else torch.zeros(M, N, device=dev))

    def run(fl):
        o = outs[fl]
        check(lib().p2w_gemm_h2(PREC, ptr(A), lda, ptr(W), 1.0, M, N, K, C.byref(ep), ptr(o) if kind == "f" else None, N,
                                ptr(o) if kind == "h" else None, ldh_o, fl, stream()))
    for fl in flagsets:
        run(fl)
    torch.cuda.synchronize()
    same = all(torch.equal(outs[fl].view(torch.int16 if kind == "h" else torch.int32),
                           outs[flagsets[0]].view(torch.int16 if kind == "h" else torch.int32)) for fl in flagsets[1:])
    t = {fl: [] for fl in flagsets}
    if os.environ.get("ONCE"):       # counter passes: two launches per flag set and shape, no timing loop
        for fl in flagsets:
            run(fl)
        torch.cuda.synchronize()
        print(f"M={M:6d} K={K:4d} N={N:4d} out={kind} same={same}", flush=True)
        continue
    for rnd in range(6):
        for fl in flagsets:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4):
                run(fl)
            e.record()
            torch.cuda.synchronize()
            t[fl].append(s.elapsed_time(e) / 4 * 1e3)
    gf = 2.0 * M * K * N / 1e9
    print(f"M={M:6d} K={K:4d} N={N:4d} out={kind} same={same}: " +
          "  ".join(f"flags{fl}: {statistics.median(t[fl]):7.1f} us ({gf / statistics.median(t[fl]) * 1e3:5.0f} TF)" for fl in flagsets),
          flush=True)
