#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/r05c_phase_cycles_wide.log; : > $L
for wd in 0 1; do
  echo "== IMCVT_HEVC_WIDE=$wd" >> $L
  IMCVT_HEVC_WIDE=$wd IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1 0 >> $L 2>&1
done
cat $L
