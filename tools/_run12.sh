cd $GRAFT_REPO_ROOT
O=gpurun_out
PP_VERBOSE=1 PP_LAUNCHES=14 timeout 400 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids | grep -v "main workgroups ran" | tail -34 | cut -c1-400 | tee $O/r03k_pace_probe.log
