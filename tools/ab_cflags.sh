#!/bin/bash
# Same-box A/B of two builds of THIS tree that differ in compile flags: tools/ab_cflags.sh "<flags A>" "<flags B>" [rounds] [bench args]
A="$1"; B="$2"; rounds=${3:-3}; shift 3
for r in $(seq 1 $rounds); do
  for v in "$A" "$B"; do
    export P2W_EXTRA_CFLAGS="$v"
    python -m pointstowood_amd.build > /dev/null || exit 1
    python bench.py --no-cpu-baseline --no-pcie --no-workloads --steps 32 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('[$v]'.ljust(34), 'ms/step %.3f median %.3f' % (d['ms_per_step'], d['ms_per_step_median']), 'gemm %.3f sa %.3f interp %.3f knn %.3f' % (k['gemm_kernel'], k['sa_conv_kernel'], k['interp_concat'], k['knn']))"
  done
done
unset P2W_EXTRA_CFLAGS
python -m pointstowood_amd.build > /dev/null
