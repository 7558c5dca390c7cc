cd $GRAFT_REPO_ROOT
O=gpurun_out
PP_VERBOSE=1 PP_LAUNCHES=16 timeout 400 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids | grep "^512\|slowest\|heartbeat" | cut -c1-900 | tee $O/r03q_hb_probe.log
