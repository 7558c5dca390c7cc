#!/bin/bash
# round 6: the context stage on table entries (block_C8e) against HEAD: A/B of the two libraries (one frame, 64 frames), timeline of the new one
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06o}
bash tools/gpu_ab_libs.sh $T head shipped | cut -c1-200
L=$O/${T}_timeline.log; : > $L
IMCVT_HEVC_PARTNERS=1 IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_tl2.so timeout 600 python tools/prof_timeline.py 1920 544 0 >> $L 2>&1
grep -n "pipe\|PU 2: decided\|PU 3: decided\|barrier\|committed\|kernel" $L | cut -c1-150
