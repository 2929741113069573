#!/bin/bash
# Round profiles (TAG, default r4) on the GPU box (run through gpurun).  Every pass is its own rocprofv3 run (PMC passes never share a run with
# each other's counter groups or with --stats, as the MI355X guide prescribes); summaries -> gpurun_out/prof_${TAG}_<name>/${TAG}_*.
#   tools/profile_round.sh bench [precision]     the bench command (BASELINE configs[1]): kernel stats (pipelined + sequential),
#                                             FETCH_SIZE / WRITE_SIZE, MFMA busy, VALU issue counters
#   tools/profile_round.sh config2|config4|surface   the named workload, 3 sequential forwards: kernel stats, FETCH / WRITE, MFMA busy
set -u
TAG=${TAG:-r6}
WHAT=${1:-bench}
PREC=${2:-f16x3}
export TMPDIR=/tmp
ROOT=$(pwd)
if [ "$WHAT" = "bench" ]; then
  NAME=$PREC; FWD=24   # forwards per sequential process: 8 allocator-sizing + 2 warmup + 10 steps + 4 of the profiled step (1 untimed + 3)
  CMD="$ROOT/bench.py --no-cpu-baseline --no-pcie --no-single-call --no-workloads --steps 10 --warmup 2 --precision $PREC"
  SEQ="--pipeline 0 --engine-opt overlap=0 --engine-opt single_res_streams=1"
else
  NAME=$WHAT; FWD=3
  CMD="$ROOT/tools/run_workload.py $WHAT 3 $PREC"
  SEQ=""
fi
OUT=$ROOT/gpurun_out/prof_${TAG}_$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp
if [ "$WHAT" = "bench" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipe -- python3 $CMD > $OUT/pipe.json 2> $OUT/pipe.err
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/seq -- python3 $CMD $SEQ > $OUT/seq.json 2> $OUT/seq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $CMD $SEQ > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $CMD $SEQ > $OUT/write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 $CMD $SEQ > $OUT/mfma.json 2> $OUT/mfma.err
if [ "$WHAT" = "bench" ]; then
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/valu -- python3 $CMD $SEQ > $OUT/valu.json 2> $OUT/valu.err
fi
cd $ROOT
python3 tools/profile_summary.py $OUT $NAME $FWD $TAG "$CMD $SEQ"
ls $OUT | head -30
