#!/usr/bin/env python3
"""In-kernel phase stamps of the 256x256 GEMM (diagnostic build -DP2W_GEMM_STAMP[=2]): per workgroup, for waves 0 and 4,
cycles of the K loop, of the per-slab barrier waits inside it, of the epilogue and (level 2) of the fragment-read waits.
Shares, not absolute times, are what to read (stamps fence the schedule)."""
import ctypes as C
import os
import statistics
import sys
os.environ.setdefault("P2W_EXTRA_CFLAGS", "-DP2W_GEMM_STAMP")

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import _lib  # noqa: E402
from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream  # noqa: E402

PREC = int(os.environ.get("PREC", "0"))
planes, ka = (2, 32) if PREC == 0 else (1, 64)
hdt = torch.bfloat16 if PREC == 2 else torch.float16
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
DBG = int(os.environ.get("DBG", "0"))     # ablation bits (needs -DP2W_GEMM_ABLATE as well)
SHAPES = [(32768, 1024, 1024), (17506, 2048, 2048), (65536, 512, 512), (65536, 128, 512), (32768, 256, 1024)]
if os.environ.get("SHAPES"):
    SHAPES = SHAPES[: int(os.environ["SHAPES"])]
for M, K, N in SHAPES:
    Np, Kp = _lib.packed_dims(N, K, PREC)
    A = torch.zeros(M, planes * Kp, device=dev, dtype=hdt)
    A[:, : planes * K] = (torch.randn(M, planes * K, device=dev, generator=g) * 0.5).to(hdt)
    W = torch.zeros(Np, planes * Kp, device=dev, dtype=hdt)     # H rows: timing only, any finite content will do
    W[:N, : planes * K] = (torch.randn(N, planes * K, device=dev, generator=g) * 0.5).to(hdt)
    bias, sc, sh = (torch.randn(N, device=dev, generator=g) for _ in range(3))
    ldh_o = (N + ka - 1) // ka * ka
    out = torch.zeros(M, planes * ldh_o, dtype=hdt, device=dev)
    stamps = torch.zeros(1024 * 2 * 8, dtype=torch.int64, device=dev)
    ep = Epilogue(ptr(bias), ptr(sc), ptr(sh), ptr(sc), ptr(stamps), None, 0, 1, 1, 0, 0)
    for _ in range(3):
        check(lib().p2w_gemm_h2(PREC, ptr(A), Kp, ptr(W), 1.0, M, N, K, C.byref(ep), None, N, ptr(out), ldh_o, 2 | (DBG << 16), stream()))
    torch.cuda.synchronize()
    st = stamps.view(1024, 2, 8).cpu()
    used = st[:, 0, 0] > 0
    n = int(used.sum())
    med = lambda t: statistics.median(t.tolist())
    for w in (0, 1):
        s_ = st[used, w]
        loop, wait, epi, rd, nslab = med(s_[:, 0]), med(s_[:, 1]), med(s_[:, 2]), med(s_[:, 3]), int(s_[0, 7])
        print(f"M={M} K={K} N={N} wave{4*w}: {n} WGs, {nslab} slabs: loop {loop:.0f} cyc ({loop/nslab:.0f}/slab), barrier wait {wait:.0f} "
              f"({100*wait/loop:.0f} %), frag wait {rd:.0f} ({100*rd/loop:.0f} %), epilogue {epi:.0f} ({100*epi/(loop+epi):.0f} % of tile)")
    t0, t1 = st[used, 0, 4], st[used, 0, 5]
    print(f"   first start {int(t0.min())} last start +{int(t0.max()-t0.min())}, tile time med {med(t1-t0):.0f} cyc, span {int(t1.max()-t0.min())} cyc", flush=True)
