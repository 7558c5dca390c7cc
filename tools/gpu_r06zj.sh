#!/bin/bash
# round 6: the shipped role rule with fuller pools (512 + 480 / 496) — and what kind of box this is (the rule of rounds 3 - 5 at 512 + 448: bimodal or flat)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06zj}
L=$O/${T}_fuller_pool_new_rule.log; : > $L
PP_VERBOSE=1 PP_LAUNCHES=10 timeout 900 python tools/pool_probe.py 1920 1080 512 0 512:480 a:a 512:496 512:480 2>&1 | grep "x 1920\|main workgroups ran" >> $L
IMCVT_POOL_ROLES_BY_ARRIVAL=1 PP_VERBOSE=1 PP_LAUNCHES=8 timeout 900 python tools/pool_probe.py 1920 1080 512 0 512:448 512:480 2>&1 | grep "x 1920\|main workgroups ran" | sed 's/^/by arrival: /' >> $L
grep "x 1920" $L | cut -c1-400; grep "main workgroups ran" $L | sort | uniq -c | sort -rn | head
