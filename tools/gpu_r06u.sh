#!/bin/bash
# round 6: how often does a pool of 992 / 1008 / 1024 workgroups (late-main take-over on) run long?  16 launches each, interleaved in blocks of 8
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06u}
PP_LAUNCHES=8 timeout 2400 python tools/pool_probe.py 1920 1080 512 0 512:480 512:496 512:480 512:496 512:464 2>&1 | grep "x 1920" > $O/${T}_pool_fill.log
cut -c1-460 $O/${T}_pool_fill.log
