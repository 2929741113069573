#!/bin/bash
# Round-2 profiles of the bench command on the GPU box (run through gpurun):
#   kernel-trace + stats (pipelined and sequential), PMC passes FETCH_SIZE / WRITE_SIZE (separate, as the MI355X guide
#   prescribes), and one pass SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES + GRBM_GUI_ACTIVE.  Summaries -> gpurun_out/prof_r2/*.{csv,json}
# usage: tools/profile_r2.sh [precision]
set -u
PREC=${1:-f16x3}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r2_$PREC
rm -rf $OUT; mkdir -p $OUT
BENCH="$ROOT/bench.py --no-cpu-baseline --no-pcie --steps 10 --warmup 2 --precision $PREC"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipe -- python3 $BENCH > $OUT/pipe.json 2> $OUT/pipe.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/seq -- python3 $BENCH --pipeline 0 > $OUT/seq.json 2> $OUT/seq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $BENCH --pipeline 0 > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $BENCH --pipeline 0 > $OUT/write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 $BENCH --pipeline 0 > $OUT/mfma.json 2> $OUT/mfma.err
cd $ROOT
python3 tools/profile_summary.py $OUT $PREC 21   # forwards per process: 8 allocator-sizing + 2 warmup + 10 steps + 1 profiled
ls -la $OUT | head -30
