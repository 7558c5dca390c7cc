#!/bin/bash
# Phase cycles per wave (IMCVT_PROF build) for one frame alone with the pipe wave off and on, then the whole GPU test suite.  usage: tools/gpu_prof_pipe.sh TAG
TAG=${1:-rXX}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm -DIMCVT_PROF imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_prof.so 2>/dev/null
for p in 0 1; do echo "== IMCVT_HEVC_PIPE=$p"; IMCVT_HEVC_PIPE=$p IMCVT_HEVC_TEAM=3 IMCVT_HEVC_LIB=$O/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1 0; done 2>&1 | grep -v amdgpu.ids > $O/${TAG}_phase_cycles_pipe.log
cat $O/${TAG}_phase_cycles_pipe.log
if [ -z "$NO_TESTS" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_gpu_tests.log; tail -4 $O/${TAG}_gpu_tests.log; fi
