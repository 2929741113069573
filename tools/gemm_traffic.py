#!/usr/bin/env python3
"""Per-launch fabric traffic of the GEMM class against one pass over each launch's operands (diagnostic).

    python tools/gemm_traffic.py gpurun_out/prof_r4_f16x3 [gpurun_out/gemm_list.json]

Takes the LAST forward of the profiled sequential bench (the profiled step: batch 0, the batch tools/gemm_list.py logs) from the
FETCH_SIZE and WRITE_SIZE passes of tools/profile_round.sh and prints, launch by launch, counter bytes (FETCH x 2 + WRITE, KiB -> B)
beside read + write bytes of one pass over A, W, residual and outputs."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
lst = json.load(open(sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/gemm_list.json"))


def gemm_rows(sub, counter):
    f = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("void gemm_hp_kernel")
            or (r["Counter_Name"] == counter and "gemm_hp_kernel" in r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


fe, wr = gemm_rows("fetch", "FETCH_SIZE"), gemm_rows("write", "WRITE_SIZE")
per = len(lst) + 1                                     # + the head operator's GEMM
fe, wr = fe[-per:], wr[-per:]
tot_c = tot_a = 0.0
print(f"{'launch':34s} {'fetch MB':>9s} {'write MB':>9s} {'one pass rd':>11s} {'wr':>7s} {'ratio':>6s}")
for i, (a, b) in enumerate(zip(fe, wr)):
    fb, wb = 2.0 * float(a["Counter_Value"]) * 1024, float(b["Counter_Value"]) * 1024
    if i < len(lst):
        g = lst[i]
        name = f"{g['name'][5:]:5s} {g['M']:6d} x {g['K']:4d} -> {g['N']:4d} {'f' if g['f32'] else ''}{'h' if g['h2'] else ''}{'+r' if g['residual'] else ''}"
        rd, w_ = g["read_bytes"], g["write_bytes"]
    else:
        name, rd, w_ = "head 131072 x 512 -> 512 . w", 131072 * 512 * 4 + 512 * 512 * 4, 131072 * 4
    tot_c += fb + wb
    tot_a += rd + w_
    print(f"{name:34s} {fb / 1e6:9.1f} {wb / 1e6:9.1f} {rd / 1e6:11.1f} {w_ / 1e6:7.1f} {(fb + wb) / (rd + w_):6.2f}")
print(f"class: counters {tot_c / 1e9:.2f} GB, one pass {tot_a / 1e9:.2f} GB, ratio {tot_c / tot_a:.2f}")
