#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=$O/r05n_clocks.log; : > $L
rocm-smi --showclocks --showperflevel >> $L 2>&1
(python tools/wide_probe.py 1920 1080 0 1 > $O/r05n_probe.log 2>&1) &
P=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk" >> $L; sleep 2; done
wait $P
echo "== default perf level" >> $L; cat $O/r05n_probe.log | grep -v amdgpu.ids >> $L
rocm-smi --setperflevel high >> $L 2>&1
rocm-smi --showclocks --showperflevel 2>&1 | grep -E "sclk|Performance" >> $L
WP_LAUNCHES=2 python tools/wide_probe.py 1920 1080 0 1 2>&1 | grep -v amdgpu.ids >> $L
rocm-smi --setperflevel auto >> $L 2>&1
cat $L
