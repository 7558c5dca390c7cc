#!/bin/bash
# round 6, second pass on the GPU: partner workgroups (tests, latency A/B), host-pointer path (follow modes), split launches with fixed halves, whole suite, bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06b}
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "partner or two_launches or follows_the_running or launches_too_large" > $O/${T}_new_tests.log 2>&1; echo "new tests rc=$?" >> $O/${T}_new_tests.log; tail -15 $O/${T}_new_tests.log
timeout 1500 python tools/r06_ab.py partners split follow --reps 3 > $O/${T}_ab.log 2>&1; echo "ab rc=$?" >> $O/${T}_ab.log; cat $O/${T}_ab.log | cut -c1-1200
timeout 2400 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/${T}_gpu_tests.log; tail -5 $O/${T}_gpu_tests.log
timeout 1200 python bench.py --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?"; tail -c 7000 $O/${T}_bench.json; tail -5 $O/${T}_bench.err
