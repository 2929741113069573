#!/usr/bin/env python3
"""Is the GEMM epilogue bound by a chip-wide resource because all workgroups store at the same time?  (diagnostic, MI355X)

Three experiments on p2w_gemm_h2 (f16x3), using the diagnostic flag bits 8..15 of `flags`:
  1. grid cap: the same launch on all / half / a quarter of the CUs (a chip-wide bottleneck makes 1/2 take less than 2x);
  2. start stagger: every other workgroup of an XCD starts X us late (integer rounds of tiles, so a pure loss of X unless
     the de-synchronised epilogues overlap the other workgroups' MFMA loops);
  3. two different layers back to back on the full chip vs side by side on two streams with half the CUs each.
"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import _lib  # noqa: E402
from pointstowood_amd._lib import Epilogue, check, lib, ptr  # noqa: E402

PREC = int(os.environ.get("PREC", "0"))
planes, ka = (2, 32) if PREC == 0 else (1, 64)
hdt = torch.bfloat16 if PREC == 2 else torch.float16
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
GDIV2, GDIV4 = 1 << 14, 1 << 15
stag = lambda us: (max(0, min(63, int(round(us / 2)))) << 8)


class Problem:
    def __init__(self, M, K, N, kind):
        self.M, self.K, self.N, self.kind = M, K, N, kind
        Np, Kp = _lib.packed_dims(N, K, PREC)
        self.lda = Kp
        self.A = torch.zeros(M, planes * Kp, device=dev, dtype=hdt)
        self.A[:, : planes * K] = (torch.randn(M, planes * K, device=dev, generator=g) * 0.5).to(hdt)
        self.W = torch.zeros(Np, planes * Kp, device=dev, dtype=hdt)
        self.W[:N, : planes * K] = (torch.randn(N, planes * K, device=dev, generator=g) * 0.5).to(hdt)
        self.bias, self.sc, self.sh = (torch.randn(N, device=dev, generator=g) for _ in range(3))
        self.ldh_o = (N + ka - 1) // ka * ka
        if kind == "h":
            self.ep = Epilogue(ptr(self.bias), ptr(self.sc), ptr(self.sh), None, None, None, 0, 1, 1, 0, 0)   # class 263 (g1)
            self.out = torch.zeros(M, planes * self.ldh_o, dtype=hdt, device=dev)
        else:
            self.ep = Epilogue(ptr(self.bias), None, None, None, None, None, 0, 1, 0, 0, 0)
            self.out = torch.zeros(M, N, device=dev)
        self.gflop = 2.0 * M * K * N / 1e9

    def run(self, flags, stream=None):
        s = (stream or torch.cuda.current_stream()).cuda_stream
        o = self.out
        check(lib().p2w_gemm_h2(PREC, ptr(self.A), self.lda, ptr(self.W), 1.0, self.M, self.N, self.K, C.byref(self.ep),
                                ptr(o) if self.kind == "f" else None, self.N, ptr(o) if self.kind == "h" else None, self.ldh_o,
                                flags, s))


def time_us(fn, reps=4, rounds=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / reps * 1e3)
    return statistics.median(ts)


print("== 1/2: grid cap and start stagger (integer rounds of 256x256 tiles) ==", flush=True)
shapes = [(131072, 128, 512, "h"), (131072, 512, 512, "h"), (131072, 512, 128, "f"),
          (65536, 256, 1024, "h"), (65536, 1024, 1024, "h"), (65536, 1024, 256, "f"),
          (131072, 512, 512, "f")]
for M, K, N, kind in shapes:
    p = Problem(M, K, N, kind)
    t0 = time_us(lambda: p.run(0))
    tiles = (M // 256) * max(1, N // 256)
    t_tile = t0 / max(1.0, tiles / 256)
    row = [f"M={M} K={K} N={N} out={kind}: full {t0:7.1f} us ({p.gflop / t0 * 1e3:4.0f} TF, ~{t_tile:5.1f} us/tile)"]
    for name, fl in (("1/2 grid", GDIV2), ("1/4 grid", GDIV4)):
        t = time_us(lambda: p.run(fl))
        row.append(f"{name} {t:7.1f} ({t / t0:4.2f}x)")
    for frac in (0.25, 0.5, 0.75):
        x = 2 * round(t_tile * frac / 2)
        t = time_us(lambda: p.run(stag(x)))
        row.append(f"stagger {x:3d} us -> {t:7.1f} ({t - t0:+6.1f})")
    print("  ".join(row), flush=True)
    del p

print("== 3: two layers, back to back on the whole chip vs side by side on half the CUs each ==", flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
pairs = [((65536, 128, 512, "h"), (65536, 512, 512, "h")), ((65536, 512, 512, "h"), (65536, 512, 128, "f")),
         ((40960, 256, 1024, "h"), (40960, 1024, 1024, "h")), ((40960, 1024, 1024, "h"), (40960, 1024, 256, "f")),
         ((65536, 512, 512, "h"), (65536, 512, 512, "h"))]
for a, b in pairs:
    pa, pb = Problem(*a), Problem(*b)

    def seq():
        pa.run(0)
        pb.run(0)

    def par(fl):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        pa.run(fl, s1)
        pb.run(fl, s2)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    t_seq = time_us(seq)
    t_par = time_us(lambda: par(GDIV2))
    t_par_full = time_us(lambda: par(0))
    print(f"{a} + {b}: sequential {t_seq:7.1f} us, two streams x half grid {t_par:7.1f} ({t_par / t_seq:4.2f}x), two streams x full grid "
          f"{t_par_full:7.1f} ({t_par_full / t_seq:4.2f}x)", flush=True)
    del pa, pb
