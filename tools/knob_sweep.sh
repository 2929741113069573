#!/bin/bash
# A/B of engine knobs on the bench forward: VAR=value pairs given as arguments, each against the default, interleaved
run() { env "$@" python bench.py --no-cpu-baseline --no-pcie --steps 16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*'.ljust(28), 'ms/step %.3f median %.3f' % (d['ms_per_step'], d['ms_per_step_median']))"; }
for r in 1 2 3; do
  run P2W_NOOP=1
  for kv in "$@"; do run $kv; done
done
