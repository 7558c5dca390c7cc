#!/bin/bash
# round 6: what is the 2x outlier of a pool that uses every slot?  512 + 512, verbose frame clocks, many launches
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06t}
PP_VERBOSE=1 PP_LAUNCHES=${2:-14} timeout 1500 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids > $O/${T}_full_pool_verbose.log
cut -c1-600 $O/${T}_full_pool_verbose.log | tail -60
