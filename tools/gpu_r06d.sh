#!/bin/bash
# round 6, fourth pass: kept PU winners in LDS (wide workgroups) — tests + latency; device layout probe (why the host-pointer path's kernel is 3.5 % slower); timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06d}
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide or partner or pipe or golden or two_launches" > $O/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O/${T}_tests.log; tail -4 $O/${T}_tests.log | cut -c1-300
timeout 1500 python tools/r06_ab.py wide layout --reps 3 > $O/${T}_ab.log 2>&1; echo "ab rc=$?" >> $O/${T}_ab.log; cat $O/${T}_ab.log | cut -c1-1200
