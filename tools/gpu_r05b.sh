#!/bin/bash
# round 5: wide workgroups (split trial coders) — parity on the device, then one frame / 64 frames with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/r05b_wide.log; : > $L
timeout 900 python tools/gpu_parity.py --big >> $L 2>&1; echo "parity rc=$?" >> $L
timeout 600 python tools/wide_probe.py 512 256 0 1 >> $L 2>&1
timeout 900 python tools/wide_probe.py 1920 1080 0 1 64 >> $L 2>&1
timeout 600 python tools/wide_probe.py 1920 1080 4 1 >> $L 2>&1
cat $L
