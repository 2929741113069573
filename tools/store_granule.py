#!/usr/bin/env python3
"""Epilogue-shaped store streams against each other (tools/micro/store_granule.hip; build first, here or on the box:
hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/store_granule.hip -o build_variants/store_granule.so):
GB/s of writing a [rows x 2048 B] matrix in 64-byte, 128-byte, 256-byte and 1-KiB pieces per row."""
import ctypes as C, os, statistics, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(root, "build_variants", "store_granule.so"))
L.store_granule.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
names = {0: "4 rows x 64 B (+ the other half next)", 1: "4 rows x 128 B", 2: "4 rows x 256 B", 3: "4 rows x 256 B as 16 B per lane, row-major lanes"}
for rows in (65536, 123046, 524288):
    out = torch.empty(rows * 2048, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for grid in (512, 2048):
        res = []
        for mode in (0, 1, 2, 3):
            for _ in range(3):
                L.store_granule(out.data_ptr(), rows, mode, grid, st)
            ts = []
            for _ in range(9):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); L.store_granule(out.data_ptr(), rows, mode, grid, st); e.record(); torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) * 1e3)
            t = statistics.median(ts)
            res.append(f"mode {mode}: {t:7.1f} us {rows * 2048 / t / 1e3:6.0f} GB/s")
        print(f"rows {rows:7d} ({rows * 2048 / 1e6:6.1f} MB) grid {grid:4d} | " + " | ".join(res), flush=True)
for k, v in names.items():
    print(f"mode {k}: {v}")
