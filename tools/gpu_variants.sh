#!/bin/bash
# Several builds of the kernel on one box, timed interleaved (1024 x 512x256, a frame per workgroup).  usage: tools/gpu_variants.sh TAG "flags A" "flags B" ...   ("" = the shipped build)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_variants.log; : > $L
n=0; libs=()
for fl in "$@"; do
  so=$O/libimcvt_hevc_v$n.so
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm $fl imcvt_amd/csrc/hevc_hip.hip -o $so 2> $O/${TAG}_v$n.build.log || echo "build failed: $fl" >> $L
  libs+=("$so"); n=$((n+1))
done
for rep in 1 2 3; do
  i=0
  for fl in "$@"; do
    echo "== [$fl]" >> $L; IMCVT_HEVC_LIB=${libs[$i]} timeout 300 python tools/quick_bench.py 512 256 1024 0 2>&1 | grep -v amdgpu.ids >> $L
    i=$((i+1))
  done
done
cat $L
