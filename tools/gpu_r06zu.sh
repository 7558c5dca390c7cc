#!/bin/bash
# round 6: do the empty launches of the re-warm, put in front of EVERY full launch, even out where a pool that uses every slot lands?  512 + 512 / 512 + 480, with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06zu}
export IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_rwalways.so
L=$O/${T}_rewarm_always.log; : > $L
for RW in 1 0; do
  echo "== IMCVT_HEVC_REWARM_ALWAYS=$RW" >> $L
  if [ $RW = 1 ]; then export IMCVT_HEVC_REWARM_ALWAYS=1; else unset IMCVT_HEVC_REWARM_ALWAYS; fi
  PP_VERBOSE=1 PP_LAUNCHES=12 timeout 900 python tools/pool_probe.py 1920 1080 512 0 512:512 512:480 2>&1 | grep "x 1920\|main workgroups ran" >> $L
done
grep "==\|x 1920" $L | cut -c1-330; grep "main workgroups ran\|==" $L | uniq -c | head -20
