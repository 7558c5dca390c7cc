cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team or pool or shape" > $O/r03g_pool_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03g_pool_tests.log
PP_VERBOSE=1 PP_LAUNCHES=10 timeout 1500 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids | tee $O/r03g_pool_probe.log
PP_LAUNCHES=3 timeout 600 python tools/pool_probe.py 1920 1080 256 0 a:a 2>&1 | grep -v amdgpu.ids | tee -a $O/r03g_pool_probe.log
PP_LAUNCHES=3 timeout 600 python tools/pool_probe.py 1920 1080 64 0 a:a 2>&1 | grep -v amdgpu.ids | tee -a $O/r03g_pool_probe.log
PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 1 0 a:a 2>&1 | grep -v amdgpu.ids | tee -a $O/r03g_pool_probe.log
PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 640 0 a:a 0:0 2>&1 | grep -v amdgpu.ids | tee -a $O/r03g_pool_probe.log
