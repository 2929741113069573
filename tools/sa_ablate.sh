#!/bin/bash
# Timing-only ablations of the fused PointNetConv kernel inside the bench forward (diagnostic build -DP2W_SA_ABLATE; results are
# wrong by construction).  dbg bits: 1 no epilogue, 2 no W2 DMA after the first, 4 no MFMA, 8 no producer, 16 no P gather.
export P2W_EXTRA_CFLAGS="-DP2W_SA_ABLATE $SA_ABLATE_EXTRA"
python -m pointstowood_amd.build > /dev/null || exit 1
for dbg in ${@:-0 2 16 8 24 18 26 27 4 1}; do
  python bench.py --no-cpu-baseline --no-pcie --no-workloads --steps 8 --warmup 2 --engine-opt sa_flags=$((dbg << 16)) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('dbg %2d' % $dbg, 'sa %.3f ms' % k['sa_conv_kernel'], ' gemm %.3f' % k['gemm_kernel'])"
done
unset P2W_EXTRA_CFLAGS
python -m pointstowood_amd.build > /dev/null
