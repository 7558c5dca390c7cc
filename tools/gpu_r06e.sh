#!/bin/bash
# round 6, fifth pass: host-pointer path after the per-stream pre-warm, split launch after its streams' pre-warm, dynamic opcode mix (region counters)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06e}
timeout 1500 python tools/r06_ab.py follow split --reps 2 > $O/${T}_ab.log 2>&1; echo "ab rc=$?" >> $O/${T}_ab.log; cat $O/${T}_ab.log | cut -c1-900
timeout 900 python tools/valu_dyn_mix.py --frames 64 --out $O/${T}_valu_dyn_mix.json --save-counts $O/${T}_region_counts.json > $O/${T}_dyn_mix.log 2>&1; echo "dyn rc=$?" >> $O/${T}_dyn_mix.log; tail -60 $O/${T}_dyn_mix.log | cut -c1-200
