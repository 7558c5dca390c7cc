#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python tools/shape64_probe.py 3 2>&1 | grep -v amdgpu.ids > $O/${1:-r06zh}_shape64.log; cut -c1-700 $O/${1:-r06zh}_shape64.log
