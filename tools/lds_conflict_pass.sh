#!/bin/bash
# One rocprofv3 --pmc pass of the sequential bench: LDS bank conflicts and MFMA busy per kernel (the "c" pass of tools/pmc_stalls.sh alone)
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/lds_pass
rm -rf $OUT; mkdir -p $OUT
CMD="$ROOT/bench.py --no-cpu-baseline --no-pcie --no-single-call --no-workloads --steps 6 --warmup 2 --pipeline 0 --engine-opt overlap=0 --engine-opt single_res_streams=1 ${BENCH_EXTRA:-}"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- python3 $CMD > $OUT/c.json 2> $OUT/c.err
cd $ROOT
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, re, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(root, "c", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:10]
for n, c in rows:
    print(f"{n:62s} lds_conflict {100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0), 1):5.1f} %  "
          f"mfma_busy {100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1024.0 * c.get('GRBM_GUI_ACTIVE', 0) / 8.0, 1):5.1f} %")
PY
