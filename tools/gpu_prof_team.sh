#!/bin/bash
# Phase cycles per role (IMCVT_PROF build) for one frame alone, by team size.  usage: tools/gpu_prof_team.sh TAG
TAG=${1:-rXX}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm -DIMCVT_PROF imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_prof.so 2>/dev/null
for t in 3 1; do IMCVT_HEVC_TEAM=$t IMCVT_HEVC_LIB=$O/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1 0; done 2>&1 | grep -v amdgpu.ids > $O/${TAG}_phase_cycles_team.log
cat $O/${TAG}_phase_cycles_team.log
