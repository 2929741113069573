#!/bin/bash
# The round's profile set in one GPU session: kernel stats (pipelined + sequential), FETCH / WRITE, MFMA busy, VALU, stall counters,
# the searches' evaluated pairs (diagnostic build), launches per forward.  Summaries land in gpurun_out/; copy them to profiles/.
set -u
export TAG=r6
bash tools/profile_round.sh bench f16x3 > gpurun_out/profile_round_r6.log 2>&1
bash tools/pmc_stalls.sh > gpurun_out/pmc_stalls_r6.log 2>&1
python tools/slab_prof.py build_variants/slabprof.so --json gpurun_out/r6_search_evaluated.json > gpurun_out/slab_prof_r6.log 2>&1
bash tools/launch_count.sh uniform 10 > gpurun_out/r6_launch_count.log 2>&1
ls gpurun_out/prof_r6_f16x3 | head -20
tail -3 gpurun_out/slab_prof_r6.log
tail -2 gpurun_out/r6_launch_count.log
