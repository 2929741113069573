#!/usr/bin/env python3
"""Does the epilogue's arithmetic cost time?  The same H-output GEMM with 0, 2, 4 and 6 affine / ReLU operations per element
(production kernel, specialised epilogues): if the times agree the epilogue is not VALU-bound (diagnostic)."""
import ctypes as C, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointstowood_amd import _lib
from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream

dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
for M, K, N in ((123046, 512, 512), (123046, 128, 512), (81683, 1024, 1024)):
    Np, Kp = _lib.packed_dims(N, K, 0)
    A = (torch.randn(M, 2 * Kp, device=dev, generator=g) * 0.5).half()
    W = (torch.randn(Np, 2 * Kp, device=dev, generator=g) * 0.5).half()
    v = [torch.randn(N, device=dev, generator=g) for _ in range(5)]
    out = torch.zeros(M, 2 * N, dtype=torch.float16, device=dev)
    eps = {"bias only (1 op)": Epilogue(ptr(v[0]), None, None, None, None, None, 0, 0, 0, 0, 0),
           "bias relu (2)": Epilogue(ptr(v[0]), None, None, None, None, None, 0, 1, 0, 0, 0),
           "bias relu affine relu (4)": Epilogue(ptr(v[0]), ptr(v[1]), ptr(v[2]), None, None, None, 0, 1, 1, 0, 0),
           "bias relu affine relu affine relu (6)": Epilogue(ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(v[3]), ptr(v[4]), None, 0, 1, 1, 1, 0)}
    t = {k: [] for k in eps}
    for rnd in range(7):
        for k, ep in eps.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4):
                check(lib().p2w_gemm_h2(0, ptr(A), Kp, ptr(W), 1.0, M, N, K, C.byref(ep), None, N, ptr(out), N, 0, stream()))
            e.record(); torch.cuda.synchronize()
            if rnd:
                t[k].append(s.elapsed_time(e) / 4 * 1e3)
    print(f"M={M} K={K} N={N}: " + "  ".join(f"{k}: {statistics.median(x):.1f} us" for k, x in t.items()), flush=True)
