#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
export IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_fullpool.so
for S in 1 0; do timeout 200 python tools/first_launches_probe.py $S 2>&1 | grep -v amdgpu.ids | tail -1; done > $O/${1:-r06zw}_first_launches.log; cut -c1-700 $O/${1:-r06zw}_first_launches.log
