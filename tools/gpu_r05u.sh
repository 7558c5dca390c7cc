#!/bin/bash
# throughput shape (512 x 1080p, 512 + 448 workgroups of 192 threads): this round's library against round 4's, interleaved on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/r05u_bench_shape_ab.log; : > $L
for rep in 1 2 3; do
  echo "== round 5" >> $L; PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids >> $L
  echo "== round 4" >> $L; IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_r04.so PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
