#!/bin/bash
# Pipe wave on one box: the pipe GPU tests, one / 64 / 256 frames of 1080p with the pipe wave off and on, and the bench shape (512 frames,
# no pipe wave) against a build of the previous commit's source (tools/_ab/libimcvt_hevc_base.so, built by hand).   usage: tools/gpu_pipe.sh TAG
TAG=${1:-r03p}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_pipe_probe.log; : > $L
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipe or small_golden or seeded or edge_cases or auto_team" > $O/${TAG}_pipe_tests.log 2>&1; echo "pytest rc=$?" | tee -a $L; tail -5 $O/${TAG}_pipe_tests.log | tee -a $L
timeout 600 python tools/pipe_probe.py 1920 1080 0 1 64 256 2>&1 | grep -v amdgpu.ids | tee -a $L
if [ -f tools/_ab/libimcvt_hevc_base.so ]; then
  for rep in 1 2; do
    echo "== this source, bench shape" | tee -a $L; PP_LAUNCHES=1 PP_MODES=0 timeout 300 python tools/pipe_probe.py 1920 1080 0 512 2>&1 | grep -v amdgpu.ids | tee -a $L
    echo "== previous commit, bench shape" | tee -a $L; IMCVT_HEVC_LIB=$R/tools/_ab/libimcvt_hevc_base.so PP_LAUNCHES=1 PP_MODES=0 timeout 300 python tools/pipe_probe.py 1920 1080 0 512 2>&1 | grep -v amdgpu.ids | tee -a $L
  done
fi
