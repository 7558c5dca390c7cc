#!/bin/bash
# round 6: the range half of the PU pricing on a wavefront of its own, following the token making: device tests of the partner path, A/B against -DPU_RANGE_WAVE=0, timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06q}; A=${2:-norange}
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "partner or wide" > $O/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O/${T}_tests.log; tail -4 $O/${T}_tests.log | cut -c1-200
bash tools/gpu_ab_libs.sh $T $A shipped | cut -c1-200
L=$O/${T}_timeline.log; : > $L
IMCVT_HEVC_PARTNERS=1 IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_tl3.so timeout 600 python tools/prof_timeline.py 1920 544 0 >> $L 2>&1
grep -n "PU 1\|pipe\|PU 2: decided\|PU 3\|barrier\|committed\|kernel" $L | cut -c1-150
