#!/bin/bash
# Phase cycles (IMCVT_PROF build) + SQ instruction mix for the current tree.  usage: tools/gpu_prof.sh TAG
TAG=${1:-rXX}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm -DIMCVT_PROF imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_prof.so 2>/dev/null
( IMCVT_HEVC_LIB=$O/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1 0;  IMCVT_HEVC_LIB=$O/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1024 0 ) 2>&1 | grep -v amdgpu.ids > $O/${TAG}_phase_cycles.log
cat $O/${TAG}_phase_cycles.log
export TMPDIR=/tmp; cd /tmp
pass() { local name=$1; shift; timeout 600 rocprofv3 --pmc "$@" -d $O/pmc_${TAG}_$name -o $name -- python $R/tools/pmc_run.py 512 256 1024 0 > $O/pmc_${TAG}_$name.log 2>&1; echo "$name rc=$?"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cd $R
for p in sq1 sq2; do d=$(find $O/pmc_${TAG}_$p -name '*.db' | head -1); [ -n "$d" ] && python tools/rocpd_pmc.py $d 131072; done 2>&1 | awk "{k=\$1; last[k]=\$0} END{for (k in last) print last[k]}" | sort > $O/${TAG}_pmc_sq.txt
cat $O/${TAG}_pmc_sq.txt
