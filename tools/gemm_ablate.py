#!/usr/bin/env python3
"""Time p2w_gemm_h2 on the network's GEMM shapes under the kernel's profiling ablations (flags bits 16..23, honoured by
-DP2W_GEMM_ABLATE builds only).  PREC=0|1|2 (f16x3 / fp16 / bf16), GEMM_FLAGS=<P2W_GEMM_* bits>, ABLATE_MODES=0,1,..."""
import ctypes as C
import os
os.environ.setdefault("P2W_EXTRA_CFLAGS", "-DP2W_GEMM_ABLATE")   # the ablation switches are compiled out of production builds
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import _lib  # noqa: E402
from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream  # noqa: E402

shapes = [(123046, 512, 512), (81683, 1024, 1024), (17506, 2048, 2048), (123046, 128, 512), (123046, 512, 128),
          (131072, 544, 512), (81683, 768, 640)]
dev = torch.device("cuda")
PREC, FLAGS = int(os.environ.get("PREC", "0")), int(os.environ.get("GEMM_FLAGS", "0"))
planes, ka = (2, 32) if PREC == 0 else (1, 64)
hdt = torch.bfloat16 if PREC == 2 else torch.float16
for M, K, N in shapes:
    Np, Kp = _lib.packed_dims(N, K, PREC)
    A = (torch.randn(M, planes * Kp, device=dev) * 0.5).to(hdt)
    W = (torch.randn(Np, planes * Kp, device=dev) * 0.5).to(hdt)
    bias = torch.randn(N, device=dev)
    sc, sh = torch.randn(N, device=dev), torch.randn(N, device=dev)
    ldh_o = (N + ka - 1) // ka * ka
    oh = torch.empty(M, planes * ldh_o, dtype=hdt, device=dev)
    of = torch.empty(M, N, device=dev)
    ep = Epilogue(ptr(bias), ptr(sc), ptr(sh), None, None, None, 0, 1, 1, 0, 0)
    res = []
    for dbg in [int(v) for v in os.environ.get("ABLATE_MODES", "0,1,2,3,4,6").split(",")]:
        for out_f, out_h in ((None, oh), (of, None)):
            def run():
                check(lib().p2w_gemm_h2(PREC, ptr(A), Kp, ptr(W), 1.0, M, N, K, C.byref(ep), ptr(out_f), N, ptr(out_h),
                                        ldh_o, FLAGS | (dbg << 16), stream()))
            run(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                run()
            e.record(); torch.cuda.synchronize()
            res.append((dbg, "h2" if out_h is not None else "f32", s.elapsed_time(e) / 5 * 1e3))
    gf = 2.0 * M * K * N / 1e9
    print(f"M={M} K={K} N={N} ({gf:.0f} GFLOP): " + "  ".join(f"dbg{d}/{o}={t:.0f}us({gf/t*1e3/1e3:.0f}TF)" for d, o, t in res), flush=True)
