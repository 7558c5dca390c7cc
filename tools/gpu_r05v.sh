#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/r05v_bench_shape_ab.log; : > $L
for rep in 1 2 3; do
  for v in cur lb256 r04; do
    echo "== $v" >> $L
    if [ $v = cur ]; then PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids >> $L
    else IMCVT_HEVC_WIDE=0 IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_$v.so PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids >> $L; fi
  done
done
cat $L
