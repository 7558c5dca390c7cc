#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06n}
timeout 3000 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/${T}_gpu_tests.log; tail -4 $O/${T}_gpu_tests.log | cut -c1-200
timeout 1500 python bench.py --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?"; tail -c 3000 $O/${T}_bench.json; tail -3 $O/${T}_bench.err
