#!/bin/bash
# instruction counts of the throughput shape: this round's library against round 4's (are the 1.3 % more instructions, or the same instructions slower?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for v in cur r04; do
  if [ $v = r04 ]; then export IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_r04.so; else unset IMCVT_HEVC_LIB; fi
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_r05x_$v -o c -- python $R/tools/pmc_run.py 512 256 1024 0 > $O/pmc_r05x_$v.log 2>&1
  d=$(find $O/pmc_r05x_$v -name '*.db' | head -1); echo "== $v"; python $R/tools/rocpd_pmc.py $d 131072 | awk '{print $1, $2, $3, $4, $6}' | sort -k1,1 -k3,3n | awk '{last[$1]=$0} END {for (k in last) print last[k]}' | sort
done 2>&1 | tee $O/r05x_pmc_insts_ab.txt
