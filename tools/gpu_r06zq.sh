#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python tools/placement_dump.py $O/${1:-r06zq}_placement.json 2>&1 | grep -v amdgpu.ids | tail -8
