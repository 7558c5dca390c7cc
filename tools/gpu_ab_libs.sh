#!/bin/bash
# A/B of library variants on one box: single 1080p frame and 64 frames, wide workgroups.  usage: tools/gpu_ab_libs.sh TAG variant...   (imcvt_amd/csrc/variants/libimcvt_hevc_<variant>.so; "shipped" = the in-tree library)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1; shift
L=$O/${T}_ab.log; : > $L
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v" >> $L
    if [ $v = shipped ]; then unset IMCVT_HEVC_LIB; else export IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_$v.so; fi
    WP_LAUNCHES=3 timeout 600 python tools/wide_probe.py 1920 1080 0 1 64 2>&1 | grep "wide 1" >> $L
  done
done
cat $L
