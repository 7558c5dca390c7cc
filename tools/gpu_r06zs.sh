#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_ldsalloc.so timeout 900 python tools/placement_dump.py $O/${1:-r06zs}_placement.json 2>&1 | grep -v amdgpu.ids | tail -8
