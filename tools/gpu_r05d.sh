#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r05d}
L=$O/${T}_wide.log; : > $L
timeout 900 python tools/gpu_parity.py --big >> $L 2>&1; echo "parity rc=$?" >> $L
timeout 900 python tools/wide_probe.py 1920 1080 0 1 64 128 >> $L 2>&1
timeout 600 python tools/wide_probe.py 1920 1080 4 1 >> $L 2>&1
P=$O/${T}_phase_cycles_wide.log; : > $P
for wd in 0 1; do
  echo "== IMCVT_HEVC_WIDE=$wd" >> $P
  IMCVT_HEVC_WIDE=$wd IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1 0 >> $P 2>&1
done
cat $L; cut -c1-330 $P
