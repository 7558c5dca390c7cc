#!/bin/bash
# The bench workload (512 x 1080p) under launch shapes with and without the pipe wave, interleaved.   usage: tools/gpu_pipe512.sh TAG
TAG=${1:-r03r}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
PP_LAUNCHES=2 timeout 900 python tools/pool_probe.py 1920 1080 512 0 a:a 512:208 512:256 a:a 512:208 512:256 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_pipe512_probe.log
