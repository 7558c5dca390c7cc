#!/bin/bash
# round 6: the role rule with a grace of 300 us: the bench shape (automatic), against the rule of rounds 3 - 5
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06ze}
L=$O/${T}_roles_default.log; : > $L
PP_VERBOSE=1 PP_OUTLIER=1 PP_BASE_MS=4650 PP_LAUNCHES=12 timeout 900 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep "x 1920\|OUTLIER: SIMD\|main workgroups ran" >> $L
IMCVT_POOL_ROLES_BY_ARRIVAL=1 PP_LAUNCHES=6 timeout 900 python tools/pool_probe.py 1920 1080 512 0 512:448 2>&1 | grep "x 1920" | sed 's/^/by arrival: /' >> $L
PP_LAUNCHES=12 timeout 900 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep "x 1920" >> $L
IMCVT_POOL_ROLES_BY_ARRIVAL=1 PP_LAUNCHES=6 timeout 900 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep "x 1920" >> $L
grep "x 1920\|OUTLIER" $L | cut -c1-420; grep "main workgroups ran" $L | sort | uniq -c
