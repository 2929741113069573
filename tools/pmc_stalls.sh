#!/bin/bash
# Where the waves of every kernel of the sequential bench step spend their cycles: three rocprofv3 --pmc passes (each its own run,
# no --stats beside them), summed per kernel by tools/pmc_stalls.py -> gpurun_out/pmc_stalls/<tag>_f16x3_stalls.csv
#   wait_pct = SQ_WAIT_ANY / SQ_WAVE_CYCLES, wait_inst = SQ_WAIT_INST_ANY / .., wait_lds = SQ_WAIT_INST_LDS / ..,
#   active: LDS / VMEM / VALU / scalar instruction-active shares, MFMA busy and MFMA-VALU co-execution, VMEM level per instruction
set -u
TAG=${TAG:-r6}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_stalls
rm -rf $OUT; mkdir -p $OUT
CMD="$ROOT/bench.py --no-cpu-baseline --no-pcie --no-single-call --no-workloads --steps 10 --warmup 2 --pipeline 0 --engine-opt overlap=0 --engine-opt single_res_streams=1"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- python3 $CMD > $OUT/a.json 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/b -- python3 $CMD > $OUT/b.json 2> $OUT/b.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- python3 $CMD > $OUT/c.json 2> $OUT/c.err
cd $ROOT
python3 tools/pmc_stalls.py $OUT $TAG
