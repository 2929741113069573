#!/bin/bash
# The MFMA-busy pass of tools/profile_round.sh alone (re-run when a pass comes back with a glitched GRBM_GUI_ACTIVE)
set -u
export TMPDIR=/tmp TAG=${TAG:-r6}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_${TAG}_f16x3
mkdir -p $OUT; rm -rf $OUT/mfma
CMD="$ROOT/bench.py --no-cpu-baseline --no-pcie --no-single-call --no-workloads --steps 10 --warmup 2 --precision f16x3 --pipeline 0 --engine-opt overlap=0 --engine-opt single_res_streams=1"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 $CMD > $OUT/mfma.json 2> $OUT/mfma.err
cd $ROOT
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, re, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
big = []
for f in glob.glob(os.path.join(root, "mfma", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "gemm_hp_kernelILi0ELi2ELi4ELi4ELi2ELb0" in k:
            big.append(float(r["Counter_Value"]))
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:8]:
    print(f"{k:62s} mfma_busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0, 1):5.1f} %")
big.sort()
print("256x256 GEMM launches' GRBM_GUI_ACTIVE: min %.3g median %.3g max %.3g (n = %d)" % (big[0], big[len(big) // 2], big[-1], len(big)))
PY
