#!/bin/bash
# round 6: what differs in the counters between the fast and the slow state of the full 192-thread launch?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; T=${1:-r06zr}
export TMPDIR=/tmp; cd /tmp
pass() { local name=$1; shift; timeout 600 rocprofv3 --pmc "$@" -d $O/pmc_${T}_$name -o $name -- python $R/tools/pmc_run_slow.py > $O/pmc_${T}_$name.log 2>&1; echo "$name rc=$?"; grep "fast\|slow" $O/pmc_${T}_$name.log | tr '\n' ' '; echo; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass lvl1 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass icache SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_IFETCH
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
cd $R
for p in sq1 sq2 lvl1 tcc icache tcp; do d=$(find $O/pmc_${T}_$p -name '*.db' | head -1); [ -n "$d" ] && python tools/rocpd_pmc.py $d 1044480; done > $O/${T}_pmc_fast_slow.txt 2>&1
rm -rf $O/pmc_${T}_*/
python - <<PY
import re, collections
rows = collections.defaultdict(list)
for l in open("$O/${T}_pmc_fast_slow.txt"):
    m = re.match(r"(\S+)\s+dispatch\s+(\d+)\s+(\d+).*lds (\d+)", l)
    if m and int(m.group(4)) == 40944 and int(m.group(3)) > 10**9: rows[m.group(1)].append(int(m.group(3)))
for k, v in rows.items():
    if len(v) >= 4: print(f"{k:30s} fast {v[0]:>16d} {v[1]:>16d}   slow {v[-2]:>16d} {v[-1]:>16d}   slow/fast {sum(v[-2:]) / max(1, sum(v[:2])):.4f}")
PY
