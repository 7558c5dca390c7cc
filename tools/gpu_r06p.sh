#!/bin/bash
# round 6: why is the kernel of the host-pointer path slower than the resident one?  (tools/host_path_probe.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06p}
timeout 900 python tools/host_path_probe.py 512 2 2>&1 | grep -v amdgpu.ids > $O/${T}_host_path.log; tail -40 $O/${T}_host_path.log | cut -c1-400
