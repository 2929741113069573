#!/usr/bin/env python3
"""What bounds the residual blocks' project layers (K = 4 F, N = F)?  p2w_gemm_h2 on H operands for growing M (A from L2 /
Infinity Cache / HBM), with and without the H residual, on the 128 x 128 tile and the library's choice: us per launch, A bytes /
time (round 5: a 256 x 128 tile on a three-stage ring - two slabs in flight - measured the same times: docs/LAB_NOTES.md).
    python tools/gemm_project_probe.py [K N]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream
L = lib()
K, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 128)
dev = torch.device("cuda")
ldh_a, ldh_o = K, N
Npad, Kpad = (N + 255) // 256 * 256, K
W = (torch.randn(Npad, 2 * Kpad, device=dev) * 0.05).half()
bias = torch.randn(N, device=dev)
print(f"K {K} N {N}: us per launch (A GB/s)")
print(f"{'M':>8s} {'A MB':>7s} | " + " | ".join(f"{n:>22s}" for n in ("128^2", "128^2 + H residual", "library + H residual")))
for M in (4096, 16384, 65536, 123046, 262144, 524288):
    A = (torch.randn(M, 2 * ldh_a, device=dev) * 0.5).half()
    R = (torch.randn(M, 2 * ldh_o, device=dev) * 0.5).half()
    out = torch.empty(M, 2 * ldh_o, dtype=torch.float16, device=dev)
    cols = []
    for flags, res in ((1, False), (1, True), (0, True)):
        ep = Epilogue(ptr(bias), None, None, None, None, ptr(R) if res else None, ldh_o if res else 0, 0, 0, 0, 1)
        f = flags | (32 if res else 0)
        run = lambda: check(L.p2w_gemm_h2(0, ptr(A), ldh_a, ptr(W), 1.0, M, N, K, C.byref(ep), None, 0, ptr(out), ldh_o, f, stream()))
        for _ in range(3):
            run()
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        t = statistics.median(ts)
        cols.append(f"{t:9.1f} ({M * 4 * K / t / 1e3:7.0f})")
    print(f"{M:8d} {M * 4 * K / 1e6:7.1f} | " + " | ".join(f"{c:>22s}" for c in cols))
