#!/bin/bash
# A/B of library variants on one box: parity of the in-tree one, then single 1080p frame + 64 frames (wide) and the bench shape (512 frames).  usage: tools/gpu_ab2.sh TAG variant...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=$1; shift
L=$O/${T}_ab.log; : > $L
timeout 900 python tools/gpu_parity.py --big 2>&1 | tail -4 >> $L
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v" >> $L
    if [ $v = shipped ]; then unset IMCVT_HEVC_LIB; else export IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_$v.so; fi
    WP_LAUNCHES=2 timeout 600 python tools/wide_probe.py 1920 1080 0 1 64 2>&1 | grep "wide 1" >> $L
    PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep "x 1920" >> $L
  done
done
unset IMCVT_HEVC_LIB
cat $L
