#!/bin/bash
# round 6: late-dispatched workgroups are slower (helpers of the last 192 blocks serve 25 % fewer requests): does it follow the scratch slot or the dispatch order?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06za}
for REV in 0 1; do
  echo "== IMCVT_SCRATCH_REV=$REV" >> $O/${T}_scratch_rev.log
  if [ $REV = 1 ]; then export IMCVT_SCRATCH_REV=1; else unset IMCVT_SCRATCH_REV; fi
  PP_VERBOSE=1 PP_OUTLIER=1 PP_BASE_MS=4700 PP_LAUNCHES=4 timeout 900 python tools/pool_probe.py 1920 1080 512 0 512:496 512:448 2>&1 | grep "helpers: mean\|mains: CTUs\|x 1920" >> $O/${T}_scratch_rev.log
done
cut -c1-330 $O/${T}_scratch_rev.log
