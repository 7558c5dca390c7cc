#!/bin/bash
# round 6: the long launches of nearly full pools: where do the slow frames run?  (tools/pool_probe.py PP_OUTLIER)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06w}
PP_VERBOSE=1 PP_OUTLIER=1 PP_BASE_MS=4700 PP_LAUNCHES=${2:-30} timeout 1500 python tools/pool_probe.py 1920 1080 512 0 ${3:-512:480} 2>&1 | grep -v amdgpu.ids > $O/${T}_outliers.log
grep "OUTLIER\|x 1920" $O/${T}_outliers.log | cut -c1-1500
