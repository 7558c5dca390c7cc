#!/bin/bash
# A/B on one box: the shipped build (out-of-line functions declared not_tail_called: their call sites lose LLVM's `tail` marker, which is
# what kept the AMDGPU backend from dropping the callee-saved register saves of internal functions) against a variant without the
# attribute (profiles/r03s2_ntc_ab.log was taken the other way round, before the attribute was adopted).   usage: tools/gpu_ntc.sh TAG
TAG=${1:-r03s2}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_ntc_ab.log; : > $L
V=$O/libimcvt_hevc_ntc.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm "-DHDN=__device__ __noinline__" imcvt_amd/csrc/hevc_hip.hip -o $V 2> $O/${TAG}_ntc.build.log || echo "variant build failed" | tee -a $L
IMCVT_HEVC_LIB=$V timeout 600 python tools/gpu_parity.py --big > $O/${TAG}_ntc_parity.log 2>&1; echo "variant parity rc=$?" | tee -a $L; tail -3 $O/${TAG}_ntc_parity.log | tee -a $L
for rep in 1 2; do
  echo "== shipped, 1024 x 512x256 solo" | tee -a $L; timeout 300 python tools/quick_bench.py 512 256 1024 0 2>&1 | grep -v amdgpu.ids | tee -a $L
  echo "== variant, 1024 x 512x256 solo" | tee -a $L; IMCVT_HEVC_LIB=$V timeout 300 python tools/quick_bench.py 512 256 1024 0 2>&1 | grep -v amdgpu.ids | tee -a $L
done
for rep in 1 2; do
  echo "== shipped, bench shape + one frame" | tee -a $L; PP_LAUNCHES=1 PP_MODES=-1 timeout 400 python tools/pipe_probe.py 1920 1080 0 512 1 2>&1 | grep -v amdgpu.ids | tee -a $L
  echo "== variant, bench shape + one frame" | tee -a $L; IMCVT_HEVC_LIB=$V PP_LAUNCHES=1 PP_MODES=-1 timeout 400 python tools/pipe_probe.py 1920 1080 0 512 1 2>&1 | grep -v amdgpu.ids | tee -a $L
done
