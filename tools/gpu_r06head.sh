#!/bin/bash
# what the driver runs at round end, on HEAD: smoke(), the GPU tests, the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06head}
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/${T}_smoke.log 2>&1; tail -2 $O/${T}_smoke.log
timeout 2400 python -m pytest tests -x -q -m gpu > $O/${T}_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/${T}_gpu_tests.log; tail -3 $O/${T}_gpu_tests.log | cut -c1-160
timeout 1500 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?"; cut -c1-400 $O/${T}_bench.json
