#!/bin/bash
# Whole GPU suite, then one / 64 frames of 1080p with the pipe wave off and on and the bench shape.   usage: tools/gpu_check.sh TAG
TAG=${1:-rXX}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_gpu_tests.log; tail -3 $O/${TAG}_gpu_tests.log
( timeout 600 python tools/pipe_probe.py 1920 1080 0 1 64; PP_LAUNCHES=2 PP_MODES=-1 timeout 400 python tools/pipe_probe.py 1920 1080 0 512 ) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_pipe_probe.log
