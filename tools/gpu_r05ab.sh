#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/r05ab_cndmask_e64_ab.log; : > $L
for rep in 1 2; do
  for v in shipped asm asmp; do
    echo "== $v" >> $L
    if [ $v = shipped ]; then unset IMCVT_HEVC_LIB; else export IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_$v.so; fi
    WP_LAUNCHES=2 timeout 600 python tools/wide_probe.py 1920 1080 0 1 2>&1 | grep "wide 1" >> $L
    PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep "x 1920" >> $L
  done
done
unset IMCVT_HEVC_LIB
IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_asmp.so timeout 900 python tools/gpu_parity.py --big 2>&1 | tail -3 >> $L
cat $L
