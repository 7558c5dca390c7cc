#!/bin/bash
# round 5, first GPU pass: what a SIMD can issue (valu_rate_probe), and the register-rich builds of the kernel on one frame
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O2 tools/valu_rate_probe.hip -o /tmp/valu_rate_probe.bin 2> $O/r05a_probe_build.log
timeout 600 /tmp/valu_rate_probe.bin $O/r05_valu_rate.json > $O/r05_valu_rate.log 2>&1; echo "probe rc=$?" >> $O/r05_valu_rate.log
L=$O/r05a_lat_variants.log; : > $L
for rep in 1 2; do
  echo "== shipped" >> $L; PP_LAUNCHES=2 timeout 300 python tools/pool_probe.py 1920 1080 1 0 a:a >> $L 2>&1
  for v in 1 2 3; do
    echo "== lat$v" >> $L; IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_lat$v.so PP_LAUNCHES=2 timeout 300 python tools/pool_probe.py 1920 1080 1 0 a:a >> $L 2>&1
  done
done
tail -60 $O/r05_valu_rate.log; cat $L
