#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06zv}
export IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_rwalways.so IMCVT_HEVC_REWARM_ALWAYS=1
PP_LAUNCHES=40 timeout 900 python tools/pool_probe.py 1920 1080 512 0 512:512 2>&1 | grep "x 1920" > $O/${T}_rewarm_always_1024.log; cut -c1-900 $O/${T}_rewarm_always_1024.log
