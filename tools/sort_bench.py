#!/usr/bin/env python3
"""p2w_sort_pairs_u64 (hand-written stable LSD radix sort, p2w_sort.h) against torch.sort(stable=True) (rocPRIM) at plot scale:
keys with 24 / 40 / 63 significant bits (cell keys of a plot: 3-4 digit passes; Morton keys: 6-8), 1 M .. 20 M pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointstowood_amd._lib import lib, ptr, stream

L = lib()
for n in (200_000, 1_000_000, 5_000_000, 19_000_000):
    for bits in (24, 40, 63):
        g = torch.Generator(device="cuda").manual_seed(n + bits)
        keys = torch.randint(0, 2 ** min(bits, 62), (n,), generator=g, dtype=torch.int64, device="cuda")
        ko, vo = torch.empty_like(keys), torch.empty(n, dtype=torch.int32, device="cuda")
        ws = torch.empty(int(L.p2w_sort_pairs_u64_ws_bytes(n)), dtype=torch.uint8, device="cuda")
        def ours():
            assert L.p2w_sort_pairs_u64(ptr(keys), ptr(ko), None, ptr(vo), n, ptr(ws), ws.numel(), stream()) == 0
        def theirs():
            return torch.sort(keys, stable=True)
        res = {}
        for name, fn in (("p2w", ours), ("torch.sort", theirs)):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / 5 * 1e3
        rk, ri = theirs()
        ok = torch.equal(ko, rk) and torch.equal(vo.long(), ri)
        print(f"n = {n:>10,d}  bits = {bits:2d}:  p2w {res['p2w']:7.3f} ms   torch.sort(stable) {res['torch.sort']:7.3f} ms   equal: {ok}", flush=True)
