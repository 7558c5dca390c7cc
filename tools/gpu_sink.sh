#!/bin/bash
# A/B on one box: the shipped build against one compiled with -mllvm -sink-insts-to-avoid-spills in addition (private segment 720 -> 528 B
# per lane): parity of the variant, full occupancy, the bench shape, one frame, HBM traffic of both (1024 x 512x256 solo).   usage: tools/gpu_sink.sh TAG
TAG=${1:-r03x2}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_sink_ab.log; : > $L
V=$O/libimcvt_hevc_sink.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills imcvt_amd/csrc/hevc_hip.hip -o $V 2> $O/${TAG}_sink.build.log || echo "variant build failed" | tee -a $L
IMCVT_HEVC_LIB=$V timeout 600 python tools/gpu_parity.py --big > $O/${TAG}_parity.log 2>&1; echo "variant parity rc=$?" | tee -a $L; tail -2 $O/${TAG}_parity.log | tee -a $L
for rep in 1 2; do
  echo "== shipped, 1024 x 512x256 solo" | tee -a $L; QB_LAUNCHES=2 timeout 300 python tools/quick_bench.py 512 256 1024 0 2>&1 | grep -v amdgpu.ids | tee -a $L
  echo "== + sink-insts-to-avoid-spills, 1024 x 512x256 solo" | tee -a $L; QB_LAUNCHES=2 IMCVT_HEVC_LIB=$V timeout 300 python tools/quick_bench.py 512 256 1024 0 2>&1 | grep -v amdgpu.ids | tee -a $L
done
for rep in 1 2; do
  echo "== shipped, bench shape + one frame" | tee -a $L; PP_LAUNCHES=1 PP_MODES=-1 timeout 400 python tools/pipe_probe.py 1920 1080 0 512 1 2>&1 | grep -v amdgpu.ids | tee -a $L
  echo "== + sink-insts-to-avoid-spills, bench shape + one frame" | tee -a $L; IMCVT_HEVC_LIB=$V PP_LAUNCHES=1 PP_MODES=-1 timeout 400 python tools/pipe_probe.py 1920 1080 0 512 1 2>&1 | grep -v amdgpu.ids | tee -a $L
done
bash tools/gpu_traffic.sh ${TAG}s > /dev/null 2>&1; IMCVT_HEVC_LIB=$V bash tools/gpu_traffic.sh ${TAG}v > /dev/null 2>&1
python -c "
import json
for t,n in (('${TAG}s','shipped'),('${TAG}v','+ sink')):
    j=json.load(open('gpurun_out/%s_pmc_traffic.json'%t)); print('traffic', n, 'MB/CTU %.3f read %.3f written %.3f'%(j['hbm_bytes_per_launch']/j['ctus']/1e6, j['hbm_read_bytes_per_launch']/j['ctus']/1e6, j['hbm_write_bytes_per_launch']/j['ctus']/1e6))
" | tee -a $L
