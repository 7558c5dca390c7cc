#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python tools/slow_process_probe5.py 2>&1 | grep -v amdgpu.ids > $O/${1:-r06zk}_slow_process5.log; tail -3 $O/${1:-r06zk}_slow_process5.log | cut -c1-700
