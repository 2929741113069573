#!/bin/bash
# rocprofv3 kernel statistics of the plot flow (BASELINE configs[3]: tools/run_plot.py --points 10000000: voxelise -> classify ->
# back-project; a 200 k-point warm-up plot first) -> gpurun_out/prof_${TAG}_plot/${TAG}_plot_kernel_stats.csv
set -u
TAG=${TAG:-r5}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_plot
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/run_plot.py --points 10000000 > $OUT/run.log 2> $OUT/run.err
cd $ROOT
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$OUT/${TAG}_plot_kernel_stats.csv" "$OUT/run.log" <<'PY'
import csv, re, sys
src, dst, log = sys.argv[1:4]
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["# rocprofv3 --kernel-trace --stats of: python3 tools/run_plot.py --points 10000000 (one 200 k-point warm-up plot + the 10 M-point plot: "
                "voxelise, classify every voxel, back-project); " + " ".join(open(log).read().split())[:400]])
    w.writerow(["Name", "Calls", "TotalDurationMs", "AverageUs", "Percentage"])
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
        name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:90]
        w.writerow([name, r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3), round(float(r["AverageNs"]) / 1e3, 2), round(100 * float(r["TotalDurationNs"]) / tot, 2)])
print(open(dst).read()[:3000])
PY
