#!/bin/bash
# round 6, last call: pools use every slot, re-warm in front of every full launch, three short real launches when a context is created — the default bench run and a slice of the GPU tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06zx}
timeout 240 python bench.py --no-cpu-baseline > $O/${T}_bench_512f.json 2> $O/${T}_bench.err; echo "bench rc=$?"; cut -c1-300 $O/${T}_bench_512f.json; tail -2 $O/${T}_bench.err
timeout 150 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_golden or device_resident or more_frames or pool_shapes or co_tenant" > $O/${T}_gpu_tests.log 2>&1; echo "tests rc=$?" >> $O/${T}_gpu_tests.log; tail -3 $O/${T}_gpu_tests.log | cut -c1-160
