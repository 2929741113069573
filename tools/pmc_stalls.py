#!/usr/bin/env python3
"""Summary of tools/pmc_stalls.sh: per kernel, sums over all launches of the three counter passes -> <tag>_f16x3_stalls.csv"""
import collections, csv, glob, os, re, sys
root, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r5")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
for sub in ("a", "b", "c"):
    seen = set()
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
            acc[n][sub + ":" + r["Counter_Name"]] += float(r["Counter_Value"])
            if sub == "a" and (r["Dispatch_Id"], n) not in seen:
                seen.add((r["Dispatch_Id"], n)); launches[n] += 1
rows = []
for n, c in acc.items():
    wc, wcb = c.get("a:SQ_WAVE_CYCLES", 0.0), c.get("b:SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    pct = lambda v, d: round(100.0 * v / d, 1) if d > 0 else ""
    rows.append([n, launches[n], f"{wc:.4g}", pct(c.get("a:SQ_WAIT_ANY", 0), wc), pct(c.get("a:SQ_WAIT_INST_ANY", 0), wc), pct(c.get("a:SQ_WAIT_INST_LDS", 0), wc),
                 pct(c.get("a:SQ_ACTIVE_INST_ANY", 0), wc), pct(c.get("b:SQ_ACTIVE_INST_VALU", 0), wcb), pct(c.get("b:SQ_ACTIVE_INST_LDS", 0), wcb),
                 pct(c.get("b:SQ_ACTIVE_INST_VMEM", 0), wcb), pct(c.get("b:SQ_ACTIVE_INST_SCA", 0), wcb),
                 pct(c.get("c:SQ_VALU_MFMA_BUSY_CYCLES", 0), 1024.0 * c.get("c:GRBM_GUI_ACTIVE", 0.0) / 8.0), pct(c.get("c:SQ_VALU_MFMA_COEXEC_CYCLES", 0), max(c.get("c:SQ_VALU_MFMA_BUSY_CYCLES", 0), 1e-9)),
                 round(c.get("c:SQ_INST_LEVEL_VMEM", 0) / max(c.get("c:SQ_INSTS_VMEM", 0), 1e-9), 1), pct(c.get("c:SQ_LDS_BANK_CONFLICT", 0), max(c.get("c:SQ_LDS_IDX_ACTIVE", 0), 1e-9))])
rows.sort(key=lambda r: -float(r[2]))
out = os.path.join(root, f"{tag}_f16x3_stalls.csv")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["# rocprofv3 --pmc, sequential bench (tools/pmc_stalls.sh); percentages of SQ_WAVE_CYCLES unless noted; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), the normalisation of *_mfma_busy.csv (SQ_BUSY_CYCLES is not per SIMD); "
                "coexec = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES; vmem_level = SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM (cycles a vector memory instruction is in flight); "
                "lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"])
    w.writerow(["kernel", "launches", "wave_cycles", "wait_any", "wait_inst_any", "wait_inst_lds", "active_any", "active_valu", "active_lds", "active_vmem", "active_scalar",
                "mfma_busy", "mfma_valu_coexec", "vmem_level_cycles", "lds_conflict"])
    w.writerows(rows[:24])
print(open(out).read())
