#!/bin/bash
# round 6: the counter passes the bench line reads, on the final library (nothing else: the GPU minutes of the round end here)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=gpurun_out; mkdir -p $R/$O; cd $R
T=${1:-r06zy}
export TMPDIR=/tmp; cd /tmp
pass() { local name=$1; shift; timeout 60 rocprofv3 --pmc "$@" -d $R/$O/pmc_${T}_$name -o $name -- python $R/tools/pmc_run.py 1920 1080 512 0 > $R/$O/pmc_${T}_$name.log 2>&1; echo "$name rc=$?"; }
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES
pass mfma SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_BUSY_CYCLES
cd $R
f=$(find $O/pmc_${T}_fetch -name '*.db' | head -1); w=$(find $O/pmc_${T}_write -name '*.db' | head -1)
[ -n "$f" ] && [ -n "$w" ] && python tools/pmc_traffic.py $f $w 512 0 1920 1080 > $O/${T}_pmc_traffic.json
for p in sq1 sq2 grbm mfma; do d=$(find $O/pmc_${T}_$p -name '*.db' | head -1); [ -n "$d" ] && python tools/rocpd_pmc.py $d 1044480; done > $O/${T}_pmc_sq.txt 2>&1
python tools/pmc_issue.py $O/${T}_pmc_sq.txt 512 1920 1080 0 "the bench's launch shape: 512 main + 512 helper workgroups" > $O/${T}_pmc_issue.json; cut -c1-300 $O/${T}_pmc_issue.json
rm -rf $O/pmc_${T}_*/
