#!/bin/bash
# A/B on one box: TU 0 of the four-TU shape of an 8x8 CU taken from the PU wave's pass over PU 0 (shipped) against a build that repeats
# the pass (-DTU0_SHARE=0): parity of the shipped build, full occupancy (1024 x 512x256 solo), the bench shape, one frame.   usage: tools/gpu_tu0.sh TAG
TAG=${1:-r03t2}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_tu0_ab.log; : > $L
V=$O/libimcvt_hevc_notu0.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm -DTU0_SHARE=0 imcvt_amd/csrc/hevc_hip.hip -o $V 2> $O/${TAG}_notu0.build.log || echo "variant build failed" | tee -a $L
timeout 600 python tools/gpu_parity.py --big > $O/${TAG}_parity.log 2>&1; echo "shipped parity rc=$?" | tee -a $L; tail -2 $O/${TAG}_parity.log | tee -a $L
echo "(pytest subset skipped in this run)" | tee -a $L
for rep in 1 2; do
  echo "== shipped, 1024 x 512x256 solo" | tee -a $L; QB_LAUNCHES=2 timeout 300 python tools/quick_bench.py 512 256 1024 0 2>&1 | grep -v amdgpu.ids | tee -a $L
  echo "== TU0_SHARE=0, 1024 x 512x256 solo" | tee -a $L; QB_LAUNCHES=2 IMCVT_HEVC_LIB=$V timeout 300 python tools/quick_bench.py 512 256 1024 0 2>&1 | grep -v amdgpu.ids | tee -a $L
done
for rep in 1 2; do
  echo "== shipped, bench shape + one frame" | tee -a $L; PP_LAUNCHES=1 PP_MODES=-1 timeout 400 python tools/pipe_probe.py 1920 1080 0 512 1 2>&1 | grep -v amdgpu.ids | tee -a $L
  echo "== TU0_SHARE=0, bench shape + one frame" | tee -a $L; IMCVT_HEVC_LIB=$V PP_LAUNCHES=1 PP_MODES=-1 timeout 400 python tools/pipe_probe.py 1920 1080 0 512 1 2>&1 | grep -v amdgpu.ids | tee -a $L
done
