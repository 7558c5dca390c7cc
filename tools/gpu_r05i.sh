#!/bin/bash
# instruction-cache behaviour of one frame alone (latency shape): is the lone wave's issue rate an instruction-fetch rate?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQ_WAIT_INST|INST_LEVEL" | head -40 > $O/r05i_avail.txt
pass() { local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" -d $O/pmc_r05i_$name -o $name -- python $R/tools/pmc_run.py 512 256 1 0 > $O/pmc_r05i_$name.log 2>&1; echo "$name rc=$?"; }
pass ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
pass ic2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES
cd $R
for p in ic1 ic2; do d=$(find $O/pmc_r05i_$p -name '*.db' | head -1); [ -n "$d" ] && python tools/rocpd_pmc.py $d 128; done > $O/r05i_pmc_icache.txt 2>&1
cat $O/r05i_avail.txt; cat $O/r05i_pmc_icache.txt; tail -3 $O/pmc_r05i_ic1.log
