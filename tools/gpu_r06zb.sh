#!/bin/bash
# round 6: main workgroups = the first blocks of the dispatch order (IMCVT_POOL_ROLES_BY_BLOCK=1) instead of the first two arrivals of every compute unit: pools that use every slot, and the plan's shape
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06zb}
export IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_rolesblk.so
L=$O/${T}_roles_by_block.log; : > $L
for RB in 1 0 1; do
  echo "== IMCVT_POOL_ROLES_BY_BLOCK=$RB" >> $L
  IMCVT_POOL_ROLES_BY_BLOCK=$RB PP_VERBOSE=1 PP_OUTLIER=1 PP_BASE_MS=4600 PP_LAUNCHES=10 timeout 900 python tools/pool_probe.py 1920 1080 512 0 512:512 512:448 2>&1 | grep "main workgroups ran\|OUTLIER: SIMD\|x 1920" >> $L
done
grep "==\|x 1920" $L | cut -c1-330; grep "main workgroups ran" $L | sort | uniq -c | sort -rn | head -12
