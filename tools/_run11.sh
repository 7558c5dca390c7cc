cd $GRAFT_REPO_ROOT
O=gpurun_out
PP_VERBOSE=1 PP_LAUNCHES=12 timeout 400 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids | grep -v "main workgroups ran" | tail -30 | cut -c1-330 | tee $O/r03j_pace_probe.log
PP_LAUNCHES=3 timeout 300 python tools/pool_probe.py 1920 1080 256 0 a:a 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300 | tee -a $O/r03j_pace_probe.log
PP_LAUNCHES=2 timeout 300 python tools/pool_probe.py 1920 1080 1000 0 a:a 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300 | tee -a $O/r03j_pace_probe.log
