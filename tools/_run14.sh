cd $GRAFT_REPO_ROOT
O=gpurun_out
PP_VERBOSE=1 PP_LAUNCHES=16 timeout 400 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids | grep -v "main workgroups ran" | grep "^512\|slowest" | cut -c1-420 | tee $O/r03m_fair_probe.log
PP_LAUNCHES=8 timeout 400 python tools/pool_probe.py 1920 1080 512 0 512:512:1000:1000:0:850:1000 512:512:16:32:0:650:1000 2>&1 | grep -v amdgpu.ids | grep "^512" | cut -c1-300 | tee -a $O/r03m_fair_probe.log
