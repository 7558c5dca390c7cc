#!/bin/bash
# round 6: main workgroups by dispatch order as the default, pool of 31/32 of the slots: device tests; the bench shape (automatic) against the rule of rounds 3 - 5 (IMCVT_POOL_ROLES_BY_ARRIVAL=1, 512 + 448); other shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06zd}
timeout 2400 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/${T}_gpu_tests.log; tail -3 $O/${T}_gpu_tests.log | cut -c1-200
L=$O/${T}_roles_default.log; : > $L
PP_LAUNCHES=12 timeout 900 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep "x 1920" >> $L
IMCVT_POOL_ROLES_BY_ARRIVAL=1 PP_LAUNCHES=8 timeout 900 python tools/pool_probe.py 1920 1080 512 0 512:448 2>&1 | grep "x 1920" | sed 's/^/by arrival: /' >> $L
PP_LAUNCHES=12 timeout 900 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep "x 1920" >> $L
PP_LAUNCHES=3 timeout 900 python tools/pool_probe.py 1920 1080 400 0 a:a 2>&1 | grep "x 1920" >> $L
IMCVT_POOL_ROLES_BY_ARRIVAL=1 PP_LAUNCHES=3 timeout 900 python tools/pool_probe.py 1920 1080 400 0 a:a 2>&1 | grep "x 1920" | sed 's/^/by arrival: /' >> $L
PP_LAUNCHES=3 timeout 900 python tools/pool_probe.py 1920 1080 256 0 a:a 2>&1 | grep "x 1920" >> $L
IMCVT_POOL_ROLES_BY_ARRIVAL=1 PP_LAUNCHES=3 timeout 900 python tools/pool_probe.py 1920 1080 256 0 a:a 2>&1 | grep "x 1920" | sed 's/^/by arrival: /' >> $L
cut -c1-420 $L
