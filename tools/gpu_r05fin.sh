#!/bin/bash
# round-5 measurement pass on one box: GPU tests, then tools/gpu_final.sh (counters on the bench shape, bench line, rocprofv3 trace of the same command, JPEG-LS, scale prediction)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r05fin}
timeout 3000 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/${T}_gpu_tests.log; tail -3 $O/${T}_gpu_tests.log
bash tools/gpu_final.sh $T 2>&1 | cut -c1-2500
