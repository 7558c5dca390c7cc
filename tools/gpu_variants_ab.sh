#!/bin/bash
# A/B of build variants of the tree's kernel on one box, interleaved: usage: tools/gpu_variants_ab.sh TAG frames "flags of variant 1" "flags of variant 2" ...
# (variant "" = the tree's defaults; "BASE" = tools/_ab/libimcvt_hevc_base.so)
TAG=$1; N=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_variants.log; : > $L
i=0; LIBS=()
for v in "$@"; do
  if [ "$v" = "BASE" ]; then LIBS+=("${BASE_LIB:-$R/tools/_ab/libimcvt_hevc_base.so}"); else
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm $v imcvt_amd/csrc/hevc_hip.hip -o $O/libv$i.so 2> $O/${TAG}_v$i.build.log || echo "variant $i build failed" >> $L
    LIBS+=("$O/libv$i.so"); fi
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for v in "$@"; do
    echo "== [$v]" >> $L; IMCVT_HEVC_LIB=${LIBS[$i]} PP_LAUNCHES=${PP_LAUNCHES:-2} timeout 900 python tools/pool_probe.py 1920 1080 $N ${QP:-0} a:a 2>&1 | grep -v amdgpu.ids >> $L
    i=$((i+1))
  done
done
cat $L
