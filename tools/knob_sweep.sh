#!/bin/bash
# A/B of engine options (EngineOptions fields) on the bench forward: key=value pairs given as arguments, each against the
# default, interleaved.  usage: tools/knob_sweep.sh res_streams=2 sampler=sort
run() { python bench.py --no-cpu-baseline --no-pcie --steps 16 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*'.ljust(36), 'ms/step %.3f median %.3f' % (d['ms_per_step'], d['ms_per_step_median']))"; }
for r in 1 2 3; do
  run
  for kv in "$@"; do run --engine-opt $kv; done
done
