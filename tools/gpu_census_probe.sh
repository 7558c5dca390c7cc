#!/bin/bash
# Does the residency census at context creation disturb later full launches?  1000 frames a frame per workgroup, three launches, with and without it.
TAG=${1:-rXX}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_census_probe.log; : > $L
for rep in 1 2; do
  echo "== census on" >> $L; PP_LAUNCHES=3 timeout 600 python tools/pool_probe.py 1920 1080 1000 0 a:a 2>&1 | grep -v amdgpu.ids >> $L
  echo "== census off" >> $L; IMCVT_HEVC_NO_CENSUS=1 PP_LAUNCHES=3 timeout 600 python tools/pool_probe.py 1920 1080 1000 0 a:a 2>&1 | grep -v amdgpu.ids >> $L
done
echo "== base" >> $L; IMCVT_HEVC_LIB=$R/tools/_ab/libimcvt_hevc_base.so PP_LAUNCHES=3 timeout 600 python tools/pool_probe.py 1920 1080 1000 0 a:a 2>&1 | grep -v amdgpu.ids >> $L
cat $L
