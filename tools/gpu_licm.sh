#!/bin/bash
# A/B on one box: the shipped build (-mllvm -disable-machine-licm: fewer loop-invariant values hoisted out of the pass loops and then
# spilled, private segment 1008 -> 720 B per lane) against one with machine LICM on: parity of the variant, full occupancy, the bench shape,
# one frame.  (profiles/r03w2_licm_ab.log was taken the other way round, before the flag was adopted.)
# usage: tools/gpu_licm.sh TAG
TAG=${1:-r03w2}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_licm_ab.log; : > $L
V=$O/libimcvt_hevc_licm.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value imcvt_amd/csrc/hevc_hip.hip -o $V 2> $O/${TAG}_licm.build.log || echo "variant build failed" | tee -a $L
IMCVT_HEVC_LIB=$V timeout 600 python tools/gpu_parity.py --big > $O/${TAG}_parity.log 2>&1; echo "variant parity rc=$?" | tee -a $L; tail -2 $O/${TAG}_parity.log | tee -a $L
for rep in 1 2; do
  echo "== shipped, 1024 x 512x256 solo" | tee -a $L; QB_LAUNCHES=2 timeout 300 python tools/quick_bench.py 512 256 1024 0 2>&1 | grep -v amdgpu.ids | tee -a $L
  echo "== machine LICM on, 1024 x 512x256 solo" | tee -a $L; QB_LAUNCHES=2 IMCVT_HEVC_LIB=$V timeout 300 python tools/quick_bench.py 512 256 1024 0 2>&1 | grep -v amdgpu.ids | tee -a $L
done
for rep in 1 2; do
  echo "== shipped, bench shape + one frame" | tee -a $L; PP_LAUNCHES=1 PP_MODES=-1 timeout 400 python tools/pipe_probe.py 1920 1080 0 512 1 2>&1 | grep -v amdgpu.ids | tee -a $L
  echo "== machine LICM on, bench shape + one frame" | tee -a $L; IMCVT_HEVC_LIB=$V PP_LAUNCHES=1 PP_MODES=-1 timeout 400 python tools/pipe_probe.py 1920 1080 0 512 1 2>&1 | grep -v amdgpu.ids | tee -a $L
done
