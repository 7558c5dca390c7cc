#!/bin/bash
# GPU test suite + phase profile of the MFMA kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r03w_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/r03w_gpu_tests.log
tail -3 $O/r03w_gpu_tests.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DIMCVT_PROF imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_prof.so 2>/dev/null
( IMCVT_HEVC_LIB=$O/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1 0;  IMCVT_HEVC_LIB=$O/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1024 0 ) 2>&1 | grep -v amdgpu.ids > $O/r03w_phase_cycles.log
cat $O/r03w_phase_cycles.log
