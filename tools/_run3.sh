cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team or pool or shape or golden" > $O/r03c_pool_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r03c_pool_tests.log
timeout 1500 python tools/pool_probe.py 1920 1080 512 0 0:0 512:512 512:512:8:16:2 512:512:4:8:0 512:512:2:4:0 512:512:16:32:0 512:512:1:1:0 512:512:1000:1000:0 2>&1 | grep -v amdgpu.ids | tee $O/r03c_pool_probe.log
timeout 600 python tools/pool_probe.py 1920 1080 256 0 a:a 256:512:1000:1000:0 2>&1 | grep -v amdgpu.ids | tee -a $O/r03c_pool_probe.log
timeout 600 python tools/pool_probe.py 1920 1080 1 0 a:a 2>&1 | grep -v amdgpu.ids | tee -a $O/r03c_pool_probe.log
timeout 900 python tools/pool_probe.py 1920 1080 1000 0 0:0 512:512 2>&1 | grep -v amdgpu.ids | tee -a $O/r03c_pool_probe.log
