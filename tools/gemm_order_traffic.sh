#!/bin/bash
# Tile-order A/B on the GEMM shapes whose W does not fit an XCD's L2: time, then FETCH_SIZE per launch (diagnostic; gpurun).
export TMPDIR=/tmp
ROOT=$(pwd)
export SHAPES=${SHAPES:-6,7,9,10,12,13,14}
python3 tools/gemm_flags_ab.py 0 8 16 2>&1 | grep -v amdgpu.ids
OUT=$ROOT/gpurun_out/order_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp
ONCE=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $ROOT/tools/gemm_flags_ab.py 0 8 16 > $OUT/log.txt 2>&1
cd $ROOT
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/order_pmc/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_hp_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
vals = [2 * float(r["Counter_Value"]) * 1024 / 1e6 for r in rows]
for i in range(0, len(vals), 6):     # per shape: flags 0, 8, 16 twice
    print("fetch MB (flags 0, 8, 16; second pass):", [round(v, 1) for v in vals[i + 3:i + 6]], " first pass:", [round(v, 1) for v in vals[i:i + 3]])
P
