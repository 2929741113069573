#!/bin/bash
# sweep of the residual / FP chain chunk budget (rows at 4F = 512) on the bench forward
for c in 32768 65536 131072 196608 262144 0; do
  for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-pcie --steps 16 --engine-opt res_chunk_rows=$c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('chunk $c', 'ms/step %.3f median %.3f' % (d['ms_per_step'], d['ms_per_step_median']), 'gemm %.3f' % k['gemm_kernel'])"
  done
done
