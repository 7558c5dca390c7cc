#!/bin/bash
# A/B on one box: tools/_ab/libimcvt_hevc_base.so (the previous round's kernel, built from its commit) against the tree's library —
# parity probe, then one 1080p frame / 64 frames / the 512-frame bench shape / 1000 frames a frame per workgroup, interleaved; then the
# phase counters of the tree's source.
# usage: tools/gpu_ab_base.sh TAG [shapes...]      (default shapes: 1 64 512 1000)
TAG=${1:-rXX}; shift; SHAPES=${@:-1 64 512 1000}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/${TAG}_ab.log; : > $L
timeout 600 python tools/gpu_parity.py --big >> $L 2>&1; echo "parity rc=$?" >> $L
B=$R/tools/_ab/libimcvt_hevc_base.so
for n in $SHAPES; do
  for rep in 1 2; do
    echo "== base, $n frames" >> $L; IMCVT_HEVC_LIB=$B timeout 600 python tools/pool_probe.py 1920 1080 $n 0 a:a 2>&1 | grep -v amdgpu.ids >> $L
    echo "== tree, $n frames" >> $L; timeout 600 python tools/pool_probe.py 1920 1080 $n 0 a:a 2>&1 | grep -v amdgpu.ids >> $L
  done
done
if [ -z "$AB_NO_PROF" ]; then
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm -DIMCVT_PROF imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_prof.so 2>/dev/null
( IMCVT_HEVC_LIB=$O/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1 0;  IMCVT_HEVC_LIB=$O/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1024 0 ) 2>&1 | grep -v amdgpu.ids > $O/${TAG}_phase_cycles.log
fi
cat $L $O/${TAG}_phase_cycles.log
