cd $GRAFT_REPO_ROOT
O=gpurun_out
IMCVT_HEVC_VERBOSE=1 timeout 300 python tools/census.py 1024 1024 2>&1 | grep -v amdgpu.ids | tee $O/r03d_census.log
timeout 600 python tools/gpu_parity.py --big 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/r03d_parity.log
PP_LAUNCHES=5 timeout 1500 python tools/pool_probe.py 1920 1080 512 0 512:512:1:1:0 512:512:1000:1000:0 512:512:4:8:0 512:512:1:1:2 0:0 2>&1 | grep -v amdgpu.ids | tee $O/r03d_pool_probe.log
PP_LAUNCHES=3 timeout 600 python tools/pool_probe.py 1920 1080 256 0 a:a 2>&1 | grep -v amdgpu.ids | tee -a $O/r03d_pool_probe.log
PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 1000 0 0:0 2>&1 | grep -v amdgpu.ids | tee -a $O/r03d_pool_probe.log
