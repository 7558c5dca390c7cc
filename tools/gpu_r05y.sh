#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for v in cur r04; do
  if [ $v = r04 ]; then export IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_r04.so; else unset IMCVT_HEVC_LIB; fi
  timeout 900 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d $O/pmc_r05y_$v -o c -- python $R/tools/pmc_run.py 1920 1080 512 0 > $O/pmc_r05y_$v.log 2>&1
  d=$(find $O/pmc_r05y_$v -name '*.db' | head -1); echo "== $v"; python $R/tools/rocpd_pmc.py $d 1044480 | awk '{print $1, $2, $3, $4, $6}' | sort -k1,1 -k3,3n | awk '{last[$1]=$0} END {for (k in last) print last[k]}' | sort; grep "kernel ms" $O/pmc_r05y_$v.log
done 2>&1 | tee $O/r05y_pmc_icache_bench_ab.txt
