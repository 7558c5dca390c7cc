#!/bin/bash
# round 6: roles by dispatch order — how full can the pool be?  976 / 992 / 1008 workgroups, 20 launches each
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06zc}
export IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_rolesblk.so
L=$O/${T}_roles_by_block_fill.log; : > $L
IMCVT_POOL_ROLES_BY_BLOCK=1 PP_VERBOSE=1 PP_OUTLIER=1 PP_BASE_MS=4600 PP_LAUNCHES=10 timeout 1500 python tools/pool_probe.py 1920 1080 512 0 512:480 512:496 512:464 512:480 512:496 2>&1 | grep "main workgroups ran\|OUTLIER: SIMD\|x 1920" >> $L
grep "x 1920" $L | cut -c1-330; grep "main workgroups ran" $L | sort | uniq -c | sort -rn | head -12
