#!/usr/bin/env python3
"""Summaries of tools/profile_round.sh's rocprofv3 runs -> small files for profiles/:
  <tag>_<name>_{pipe,seq}_kernel_stats.csv   rocprofv3 --stats per-kernel table (names shortened)
  <tag>_<name>_hbm_traffic.json              FETCH_SIZE (x2: gfx950 counts wide reads at half their bytes) and WRITE_SIZE per kernel
                                             class and step, launches per step -> what bench.py reads for roofline.traffic
  <tag>_<name>_mfma_busy.csv                 SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE per kernel
  <tag>_<name>_valu.csv                      SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES / SQ_BUSY_CYCLES per kernel
usage: profile_summary.py <dir> <name: precision or workload> <forwards per process> [tag=r3] [profiled command]"""
import collections
import csv
import glob
import json
import os
import re
import sys

root, prec, forwards = sys.argv[1], sys.argv[2], float(sys.argv[3])
tag = sys.argv[4] if len(sys.argv) > 4 else "r3"
command = sys.argv[5] if len(sys.argv) > 5 else ""


def short(name):
    n = re.sub(r"\(.*", "", name).replace("void ", "")
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", n)
    m = re.match(r"_Z\d+(gemm_h2g_kernel|gemm_hp_kernel|sa_conv16p_kernel)ILi(\d)ELi(\d+)ELi(\d+)E(?:Li(\d+)E)?(?:Li(\d+)E)?(?:Lb(\d)E)?", n)
    if m:
        if m.group(1) in ("gemm_h2g_kernel", "gemm_hp_kernel"):
            return f"{m.group(1)}<prec{m.group(2)},{m.group(3)},{m.group(4)},{m.group(5)},{m.group(6)}" + (",rowdot>" if m.group(7) == "1" else ">")
        return f"sa_conv16p_kernel<prec{m.group(2)},{m.group(3)},{m.group(4)},G{m.group(5)}>"
    m = re.match(r"_Z\d+(gemm_hp64_kernel|gemm_hp_sk_kernel)ILi(\d)E", n)
    if m:
        return f"{m.group(1)}<prec{m.group(2)}>" + (" (64x128 tile)" if m.group(1) == "gemm_hp64_kernel" else " (split-K pieces)")
    return n[:90]


def klass(name):
    if "gemm_h2g_kernel" in name or "gemm_hp" in name or name.startswith("gemm_kernel") or "rowdot_finish" in name:
        return "gemm_kernel"
    if "sa_conv16p_kernel" in name or "sa_conv16s_kernel" in name or "sa_edge_meta" in name or "sa_part_" in name or name.startswith("sa_conv_kernel"):
        return "sa_conv_kernel"
    for k in ("interp_concat", "segment_max", "level_gather", "rowdot", "stem_kernel", "concat_xyz", "slab_search", "knn_hint"):
        if k in name:
            return k
    if name.startswith(("vs_", "tk_", "rs_", "xs_")) or "rocprim" in name:
        return "voxel_sample"
    return "other"


for mode in ("pipe", "seq"):
    files = glob.glob(os.path.join(root, mode, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    # the pipelined bench sizes the allocator pools with one forward more than the sequential one (bench.py, setup)
    fwd_mode = forwards + (1 if (mode == "pipe" and "bench.py" in command) else 0)
    with open(os.path.join(root, f"{tag}_{prec}_{mode}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        cmd = re.sub(r"\S*/(bench\.py|tools/run_workload\.py)", r"\1", command).replace(" --pipeline 0", "").replace(" --engine-opt overlap=0 --engine-opt single_res_streams=1", "") or "bench.py"
        w.writerow([f"# rocprofv3 --kernel-trace --stats of: python3 {cmd}" + (" --pipeline 0 --engine-opt overlap=0 --engine-opt single_res_streams=1" if (mode == "seq" and "bench.py" in cmd) else "")
                    + f" ({int(fwd_mode)} forwards per process)"])
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "per_forward_us"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"],
                        f"{float(r['TotalDurationNs']) / fwd_mode / 1e3:.1f}"])


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    return acc, n


fa, fn = counters("fetch")
wa, wn = counters("write")
if fa or wa:
    per = collections.defaultdict(lambda: {"fetch_bytes_per_step": 0.0, "write_bytes_per_step": 0.0, "launches_per_step": 0.0})
    kernels = {}
    for k in set(fa) | set(wa):
        fetch = 2.0 * fa.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 / forwards      # KiB, and x2 on gfx950
        write = wa.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / forwards
        launches = max(fn.get((k, "FETCH_SIZE"), 0), wn.get((k, "WRITE_SIZE"), 0)) / forwards
        kernels[k] = {"fetch_bytes_per_step": fetch, "write_bytes_per_step": write, "launches_per_step": launches}
        c = per[klass(k)]
        c["fetch_bytes_per_step"] += fetch; c["write_bytes_per_step"] += write; c["launches_per_step"] += launches
    json.dump({"precision": prec, "forwards_per_process": forwards,
               "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of the sequential bench; "
                         "FETCH_SIZE x 2 (gfx950 counts wide coalesced reads at half their bytes, MI355X_MICROARCH.md); KiB -> bytes; "
                         "these are L2 fabric-port requests, Infinity-Cache hits included",
               "kernels": dict(per), "by_kernel": kernels},
              open(os.path.join(root, f"{tag}_{prec}_hbm_traffic.json"), "w"), indent=1, sort_keys=True)

ma, mn = counters("mfma")
if ma:
    with open(os.path.join(root, f"{tag}_{prec}_mfma_busy.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, sequential bench; sums over "
                    "all launches of a kernel.  mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"])
        w.writerow(["kernel", "launches", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "mfma_busy_pct"])
        for k in sorted(ma, key=lambda k: -ma[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
            c = ma[k]
            gui = c.get("GRBM_GUI_ACTIVE", 0.0)
            pct = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui / 8.0) if gui else 0.0
            w.writerow([k, mn.get((k, "GRBM_GUI_ACTIVE"), 0), f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.0f}", f"{c.get('SQ_BUSY_CYCLES', 0):.0f}",
                        f"{gui:.0f}", f"{pct:.1f}"])
va, vn = counters("valu")
if va:
    with open(os.path.join(root, f"{tag}_{prec}_valu.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, sequential "
                    "bench; sums over all launches of a kernel.  valu_per_wave_cycle = SQ_INSTS_VALU / (4 x SQ_WAVE_CYCLES) (the SQ counts "
                    "quad-cycles); valu_active_pct = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; us_per_step from GRBM_GUI_ACTIVE is not given: "
                    "see the kernel stats"])
        w.writerow(["kernel", "launches", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE",
                    "valu_insts_per_step", "valu_active_pct_of_wave_cycles"])
        for k in sorted(va, key=lambda k: -va[k].get("SQ_INSTS_VALU", 0)):
            c = va[k]
            wc = c.get("SQ_WAVE_CYCLES", 0.0)
            w.writerow([k, vn.get((k, "SQ_INSTS_VALU"), 0), f"{c.get('SQ_INSTS_VALU', 0):.0f}", f"{c.get('SQ_ACTIVE_INST_VALU', 0):.0f}",
                        f"{wc:.0f}", f"{c.get('SQ_BUSY_CYCLES', 0):.0f}", f"{c.get('GRBM_GUI_ACTIVE', 0):.0f}",
                        f"{c.get('SQ_INSTS_VALU', 0) / forwards:.0f}", f"{100.0 * c.get('SQ_ACTIVE_INST_VALU', 0) / wc:.1f}" if wc else ""])
print("summaries written to", root)
