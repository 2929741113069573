#!/usr/bin/env python3
"""What could ONE grouped launch for the two feature phases in flight gain on the narrow GEMM layers?  

Net.stream keeps two feature phases in flight on two streams; their narrow layers (expand K = F, project N = F, hoists) have the
same (N, K) and W.  A grouped launch would run both batches' row tiles as one persistent launch that re-uses the W slabs.  Its
upper bound is measurable without building it: time, per shape,
  (a) one launch alone (M rows),
  (b) two launches of M rows on two streams at once (what the pipeline does today; time until both are done),
  (c) ONE launch over 2 M rows with the same W (what a grouped launch would be, minus nothing).
If (c) is not below (b) the grouped launch has nothing to give.     python tools/gemm_group_probe.py
"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import _lib  # noqa: E402
from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream  # noqa: E402

PREC, planes, ka, hdt = 0, 2, 32, torch.float16
dev = torch.device("cuda")
shapes = [(131072, 32, 64, "f"), (123046, 128, 192, "f"), (81683, 256, 384, "f"),      # hoists
          (123046, 128, 512, "h"), (123046, 512, 128, "f"),                            # expand / project, level 1
          (81683, 256, 1024, "h"), (81683, 1024, 256, "f"),                            # level 2
          (17506, 512, 2048, "h"), (17506, 2048, 512, "f"),                            # level 3
          (123046, 512, 512, "h"), (81683, 1024, 1024, "h")]                           # (fat layers, for comparison)
g = torch.Generator(device="cuda").manual_seed(0)
s1, s2 = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)
print("shape: (a) one launch | (b) two launches on two streams | (c) one launch over 2 M rows   [us]")
for M, K, N, kind in shapes:
    Np, Kp = _lib.packed_dims(N, K, PREC)
    A = torch.zeros(2 * M, planes * Kp, device=dev, dtype=hdt)
    A[:, : planes * K] = (torch.randn(2 * M, planes * K, device=dev, generator=g) * 0.5).to(hdt)
    W = torch.zeros(Np, planes * Kp, device=dev, dtype=hdt)
    W[:N, : planes * K] = (torch.randn(N, planes * K, device=dev, generator=g) * 0.5).to(hdt)
    bias, sc, sh = (torch.randn(N, device=dev, generator=g) for _ in range(3))
    ldh_o = (N + ka - 1) // ka * ka
    ep = Epilogue(ptr(bias), ptr(sc), ptr(sh), None, None, None, 0, 1, 1, 0, 0) if kind == "h" else \
        Epilogue(ptr(bias), None, None, None, None, None, 0, 1, 0, 0, 0)
    out = torch.zeros(2 * M, planes * ldh_o, dtype=hdt, device=dev) if kind == "h" else torch.zeros(2 * M, N, device=dev)
    row_a, row_o = planes * Kp, (planes * ldh_o if kind == "h" else N)

    def run(r0, m):
        a, o = A[r0:], out[r0:]
        check(lib().p2w_gemm_h2(PREC, ptr(a), Kp, ptr(W), 1.0, m, N, K, C.byref(ep), ptr(o) if kind == "f" else None, N,
                                ptr(o) if kind == "h" else None, ldh_o, 0, stream()))

    def timed(fn):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3

    def one():
        for _ in range(4):
            run(0, M)

    def two_streams():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record()
        for st, r0 in ((s1, 0), (s2, M)):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                for _ in range(4):
                    run(r0, M)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    def grouped():
        for _ in range(4):
            run(0, 2 * M)
    for fn in (one, two_streams, grouped):
        fn()
    ta, tb, tc = ([timed(fn) / 4 for _ in range(7)] for fn in (one, two_streams, grouped))
    a, b, c = (statistics.median(t) for t in (ta, tb, tc))
    print(f"M={M:6d} K={K:4d} N={N:4d}: (a) {a:7.1f}  (b) {b:7.1f} = {b / a:4.2f} a  (c) {c:7.1f} = {c / a:4.2f} a   grouped vs two streams: {100 * (c / b - 1):+5.1f} %", flush=True)
