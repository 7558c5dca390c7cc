#!/bin/bash
# A/B on one box: the shipped kernel against a variant built with extra flags (parity, 1024 x 512x256 solo probe, the bench shape).
# usage: tools/gpu_ab.sh TAG flags-of-the-variant...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_ab.log; : > $L
timeout 600 python tools/gpu_parity.py --big >> $L 2>&1; echo "parity rc=$?" >> $L
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm "$@" imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_var.so 2> $O/${TAG}_var.build.log || echo "variant build failed" >> $L
for rep in 1 2; do
  echo "== shipped" >> $L; timeout 300 python tools/quick_bench.py 512 256 1024 0 >> $L 2>&1
  echo "== variant $*" >> $L; IMCVT_HEVC_LIB=$O/libimcvt_hevc_var.so timeout 300 python tools/quick_bench.py 512 256 1024 0 >> $L 2>&1
done
if [ -z "$AB_SHORT" ]; then
echo "== shipped, bench shape" >> $L; timeout 600 python tools/pool_probe.py 1920 1080 512 0 a:a >> $L 2>&1
echo "== variant, bench shape" >> $L; IMCVT_HEVC_LIB=$O/libimcvt_hevc_var.so timeout 600 python tools/pool_probe.py 1920 1080 512 0 a:a >> $L 2>&1
echo "== shipped, one frame" >> $L; timeout 600 python tools/pool_probe.py 1920 1080 1 0 a:a >> $L 2>&1
fi
cat $L
