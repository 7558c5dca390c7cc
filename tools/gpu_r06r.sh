#!/bin/bash
# round 6: does a fuller pool (512 + 480 / 496 / 512 workgroups instead of 512 + 448) still buy throughput on the bench shape?  interleaved with the automatic shape
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06r}
PP_LAUNCHES=2 timeout 1500 python tools/pool_probe.py 1920 1080 512 0 a:a 512:480 512:496 a:a 512:512 512:480 a:a 512:496 2>&1 | grep "x 1920" > $O/${T}_fuller_pool.log
cut -c1-230 $O/${T}_fuller_pool.log
