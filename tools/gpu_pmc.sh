#!/bin/bash
# Counter passes (each its own rocprofv3 run, --pmc only): HBM traffic + SQ issue/stall breakdown.  usage: tools/gpu_pmc.sh TAG [w h frames q]
TAG=${1:-rXX}; W=${2:-512}; H=${3:-256}; N=${4:-1024}; Q=${5:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
pass() {  # name counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" -d $O/pmc_${TAG}_$name -o $name -- python $R/tools/pmc_run.py $W $H $N $Q > $O/pmc_${TAG}_$name.log 2>&1
  echo "$name rc=$?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass grbm GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES
pass mfma SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_BUSY_CYCLES
# what the parked wave-cycles wait for: average VMEM / LDS instructions in flight -> latency = level / count; L2 hits and misses
pass lvl1 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES
pass lvl2 SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass icache SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_IFETCH
cd $R
f=$(find $O/pmc_${TAG}_fetch -name '*.db' | head -1); w=$(find $O/pmc_${TAG}_write -name '*.db' | head -1)
python tools/pmc_traffic.py $f $w $N $Q $W $H > $O/${TAG}_pmc_traffic.json; cat $O/${TAG}_pmc_traffic.json
NCTU=$(( (W+31)/32 * ((H+31)/32) * N ))
for p in sq1 sq2 grbm mfma lvl1 lvl2 tcc icache; do d=$(find $O/pmc_${TAG}_$p -name '*.db' | head -1); [ -n "$d" ] && python tools/rocpd_pmc.py $d $NCTU; done > $O/${TAG}_pmc_sq.txt 2>&1
cat $O/${TAG}_pmc_sq.txt
