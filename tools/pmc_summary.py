#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection CSVs -> one line per dispatch (or mean per kernel with --mean)."""
import csv, glob, sys, collections
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
mean = "--mean" in sys.argv
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
disp = collections.OrderedDict()
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat and pat not in k:
            continue
        key = (int(r["Dispatch_Id"]), k.split("(")[0][:40], r.get("Grid_Size", ""))
        disp.setdefault(key, {})[r["Counter_Name"]] = disp.get(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
keys = sorted(disp)
if last:
    keys = keys[-last:]
if mean:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for k in keys:
        cnt[k[1:]] += 1
        for c, v in disp[k].items():
            acc[k[1:]][c] += v
    for k, cs in acc.items():
        print(k[0], "grid", k[1], " ".join(f"{c}={v/cnt[k]:.4g}" for c, v in sorted(cs.items())), "n=", cnt[k])
else:
    for k in keys:
        print(k[0], k[1], " ".join(f"{c}={v:.4g}" for c, v in sorted(disp[k].items())))
