cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team or pool or shape" > $O/r03b_pool_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03b_pool_tests.log
timeout 600 python tools/pool_probe.py 1920 1080 1 0 a:a 1:1 0:0 2>&1 | grep -v amdgpu.ids | tee $O/r03b_pool_probe.log
timeout 900 python tools/pool_probe.py 1920 1080 512 0 512:512 256:512 0:0 448:576 2>&1 | grep -v amdgpu.ids | tee -a $O/r03b_pool_probe.log
timeout 600 python tools/pool_probe.py 1920 1080 256 0 a:a 256:256 2>&1 | grep -v amdgpu.ids | tee -a $O/r03b_pool_probe.log
timeout 600 python tools/pool_probe.py 1920 1080 64 0 a:a 2>&1 | grep -v amdgpu.ids | tee -a $O/r03b_pool_probe.log
