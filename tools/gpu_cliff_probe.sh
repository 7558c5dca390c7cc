#!/bin/bash
# How often does a launch that nearly fills the device start a workgroup late?  n frames a frame per workgroup (n > 640), PP_LAUNCHES launches each.
TAG=${1:-rXX}; shift; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_cliff_probe.log; : > $L
for n in ${@:-944 976 1000 1016}; do PP_LAUNCHES=${PP_LAUNCHES:-6} timeout 900 python tools/pool_probe.py 1920 1080 $n 0 a:a 2>&1 | grep -v amdgpu.ids >> $L; done
cat $L
