#!/bin/bash
# round 6, third pass: timeline of an 8x8 CU with a partner workgroup (main + partner), solo 1000-frame residency with / without the wide instantiation, host path kernel time
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06c}
L=$O/${T}_timeline.log; : > $L
for P in 1 0; do
  echo "== IMCVT_HEVC_PARTNERS=$P" >> $L
  IMCVT_HEVC_PARTNERS=$P IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_tl.so timeout 600 python tools/prof_timeline.py 1920 544 0 >> $L 2>&1
done
cat $L | cut -c1-200
timeout 1500 python tools/r06_ab.py solo follow --reps 2 > $O/${T}_ab.log 2>&1; echo "ab rc=$?" >> $O/${T}_ab.log; cat $O/${T}_ab.log | cut -c1-1200
