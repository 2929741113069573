#!/bin/bash
# Same-box A/B of two source trees (this one vs ab_old/): alternate short bench runs, print ms/step and the kernel classes.
# usage: tools/ab_bench.sh [rounds] [extra bench args for both]     (ab_old/ = `git archive <commit> | tar -x -C ab_old`, built;
# a TEMPORARY export made just before the gpurun call - it is git-ignored and pytest-ignored - and deleted after it.  For kernel
# variants of the SAME tree prefer tools/build_variant.sh + the *_ab.py tools: no second tree needed)
rounds=${1:-3}; shift
for r in $(seq 1 $rounds); do
  for t in . ab_old; do
    extra=""; grep -q -- "--no-workloads" $t/bench.py && extra="--no-workloads"
    (cd $t && python bench.py --no-cpu-baseline --no-pcie $extra --steps 24 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$t'.ljust(7), 'ms/step %.3f median %.3f' % (d['ms_per_step'], d.get('ms_per_step_median', 0)), 'gemm %.3f sa %.3f interp %.3f rowdot %.3f' % (k['gemm_kernel'], k['sa_conv_kernel'], k['interp_concat'], k.get('rowdot', 0)))")
  done
done
