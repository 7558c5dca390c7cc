#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06j}
timeout 1200 python tools/r06_ab.py split2 --reps 3 > $O/${T}_ab.log 2>&1; echo "ab rc=$?" >> $O/${T}_ab.log; cat $O/${T}_ab.log | cut -c1-900
for H in 3 2; do IMCVT_HEVC_SPLIT_HPC=$H timeout 1200 python tools/r06_ab.py split2 --reps 2 2>&1 | grep probe | cut -c1-600; done >> $O/${T}_ab.log; tail -2 $O/${T}_ab.log | cut -c1-600
