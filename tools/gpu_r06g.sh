#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06g}
timeout 900 python tools/valu_dyn_mix.py --frames 64 --out $O/${T}_valu_dyn_mix.json --save-counts $O/${T}_region_counts.json > $O/${T}_dyn_mix.log 2>&1; echo "dyn rc=$?" >> $O/${T}_dyn_mix.log; grep -i "flushes\|share_in\|mix_weighted_cycles_simd\|rc=" $O/${T}_dyn_mix.log | cut -c1-200
