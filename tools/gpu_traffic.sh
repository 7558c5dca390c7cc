#!/bin/bash
# HBM traffic counters only (two rocprofv3 --pmc passes, calibrated by a 1 GiB copy).  usage: tools/gpu_traffic.sh TAG [w h frames q]
TAG=${1:-rXX}; W=${2:-512}; H=${3:-256}; N=${4:-1024}; Q=${5:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/pmc_${TAG}_$c -o $c -- python $R/tools/pmc_run.py $W $H $N $Q > $O/pmc_${TAG}_$c.log 2>&1; echo "$c rc=$?"
done
cd $R
f=$(find $O/pmc_${TAG}_FETCH_SIZE -name '*.db' | head -1); w=$(find $O/pmc_${TAG}_WRITE_SIZE -name '*.db' | head -1)
python tools/pmc_traffic.py $f $w $N $Q $W $H > $O/${TAG}_pmc_traffic.json; cat $O/${TAG}_pmc_traffic.json
