#!/bin/bash
# round 6: pools that use every workgroup slot (a running helper takes a main index that is still free when the launch is 2 ms old): device tests, then the bench shape — automatic (512 + 512) against 512 + 448, six launches each
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06s}
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rare_paths or pool or launches or co_tenant or more_frames" > $O/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O/${T}_tests.log; tail -4 $O/${T}_tests.log | cut -c1-200
PP_LAUNCHES=6 timeout 1500 python tools/pool_probe.py 1920 1080 512 0 a:a 512:448 a:a 2>&1 | grep "x 1920" > $O/${T}_full_pool.log
cut -c1-400 $O/${T}_full_pool.log
