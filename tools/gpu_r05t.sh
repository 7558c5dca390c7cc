#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r05t}
timeout 3000 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/${T}_gpu_tests.log
timeout 1500 python bench.py > $O/${T}_bench_512f.json 2> $O/${T}_bench.err; echo "bench rc=$?" >> $O/${T}_bench.err
timeout 1500 python tools/scale_predict.py --out $O/${T}_scale_prediction.json > $O/${T}_scale.log 2>&1
tail -5 $O/${T}_gpu_tests.log; tail -3 $O/${T}_bench.err; tail -3 $O/${T}_scale.log | cut -c1-1500
