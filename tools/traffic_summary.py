#!/usr/bin/env python3
"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> per-kernel KiB per forward (gfx950: fetch doubled)."""
import csv, glob, sys, collections, re
fdir, wdir, forwards = sys.argv[1], sys.argv[2], float(sys.argv[3])
def load(d, counter):
    acc, n = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            k = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", k)[:70]
            acc[k] += float(r["Counter_Value"]); n[k] += 1
    return acc, n
fa, fn = load(fdir, "FETCH_SIZE")
wa, wn = load(wdir, "WRITE_SIZE")
print("kernel,launches_per_forward,FETCH_SIZE_KiB,fetch_corrected_MB,WRITE_SIZE_KiB,write_MB")
tf = tw = 0.0
for k in sorted(fa, key=lambda k: -(2 * fa[k] + wa.get(k, 0))):
    f, w = fa[k] / forwards, wa.get(k, 0.0) / forwards
    if 2 * f + w < 1024:
        continue
    tf += 2 * f * 1024 / 1e6; tw += w * 1024 / 1e6
    print(f"{k},{fn[k] / forwards:.1f},{f:.0f},{2 * f * 1024 / 1e6:.1f},{w:.0f},{w * 1024 / 1e6:.1f}")
print(f"# total per forward: fetch_corrected {tf:.0f} MB, write {tw:.0f} MB")
