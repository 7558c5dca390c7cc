#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/r05w_bench_shape_ab.log; : > $L
for rep in 1 2 3; do
  echo "== round 5" >> $L; PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids >> $L
  echo "== round 4" >> $L; IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_r04.so PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids >> $L
done
timeout 900 python tools/wide_probe.py 1920 1080 0 1 64 >> $L 2>&1
cat $L
