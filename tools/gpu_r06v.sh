#!/bin/bash
# round 6: the long launches of nearly full pools whose workgroups all started on time: 512 + 480, verbose frame clocks
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06v}
PP_VERBOSE=1 PP_LAUNCHES=${2:-20} timeout 1500 python tools/pool_probe.py 1920 1080 512 0 512:480 2>&1 | grep -v amdgpu.ids > $O/${T}_992_verbose.log
grep "launch \|x 1920" $O/${T}_992_verbose.log | cut -c1-420
