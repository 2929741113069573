#!/bin/bash
# Launches per forward by kernel name: rocprofv3 --kernel-trace --stats of N sequential forwards (tools/run_phase.py-like loop)
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/launch_count
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/forward_loop.py ${1:-uniform} ${2:-10} > $OUT/run.log 2> $OUT/run.err
cd $ROOT
python3 - "$OUT" "${2:-10}" <<'PY'
import csv, glob, os, re, sys
root, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(os.path.join(root, "t", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in sorted(rows, key=lambda r: -int(r["Calls"])):
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:70]
    calls = int(r["Calls"])
    tot += calls
    print(f"{name:72s} {calls / (n + 3):6.1f} per forward  avg {float(r['AverageNs']) / 1e3:7.1f} us")
print("total launches per forward (incl. 3 setup forwards in the divisor):", round(tot / (n + 3), 1))
PY
