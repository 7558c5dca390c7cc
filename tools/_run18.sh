cd $GRAFT_REPO_ROOT
O=gpurun_out
PP_VERBOSE=1 PP_LAUNCHES=20 timeout 500 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids | grep "^512\|slowest\|heartbeat" | cut -c1-600 | tee $O/r03r_probe.log
PP_VERBOSE=1 PP_LAUNCHES=10 timeout 500 python tools/pool_probe.py 1920 1080 512 0 512:512 2>&1 | grep -v amdgpu.ids | grep "^512\|slowest\|heartbeat" | cut -c1-600 | tee -a $O/r03r_probe.log
