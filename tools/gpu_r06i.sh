#!/bin/bash
# round 6: which roles of a wide workgroup's 8x8 CU share a SIMD — interleaved A/B of four placements (one 1080p frame, 64 frames)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06i}
L=$O/${T}_role_perm_ab.log; : > $L
for rep in 1 2 3; do
  for v in id v1 v2 v3; do
    echo "== $v" >> $L
    IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_perm_$v.so WP_LAUNCHES=2 timeout 600 python tools/wide_probe.py 1920 1080 0 1 64 2>&1 | grep "wide 1" >> $L
  done
done
cat $L | cut -c1-200
