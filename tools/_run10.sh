cd $GRAFT_REPO_ROOT
O=gpurun_out
PP_VERBOSE=1 PP_LAUNCHES=12 timeout 400 python tools/pool_probe.py 1920 1080 512 0 512:512:1000:1000:0:850:1000 2>&1 | grep -v amdgpu.ids | tail -30 | cut -c1-300 | tee $O/r03i_hang_probe.log
PP_LAUNCHES=6 timeout 300 python tools/pool_probe.py 1920 1080 512 0 512:512:1000:1000:0:1000:1000 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-300 | tee -a $O/r03i_hang_probe.log
