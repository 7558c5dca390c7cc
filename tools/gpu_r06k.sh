#!/bin/bash
# round 6: partner workgroups, second version (candidates keep their reconstruction, the four-TU set's tokens on a wavefront of their own): tests, A/B, timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06k}
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "partner or launches_too_large" > $O/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O/${T}_tests.log; tail -4 $O/${T}_tests.log | cut -c1-200
timeout 1200 python tools/r06_ab.py partners --reps 3 > $O/${T}_ab.log 2>&1; echo "ab rc=$?" >> $O/${T}_ab.log; cat $O/${T}_ab.log | cut -c1-600
L=$O/${T}_timeline.log; : > $L
for P in 1 0; do
echo "== IMCVT_HEVC_PARTNERS=$P" >> $L
IMCVT_HEVC_PARTNERS=$P IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_tl.so timeout 600 python tools/prof_timeline.py 1920 544 0 >> $L 2>&1
done
cat $L | cut -c1-160
