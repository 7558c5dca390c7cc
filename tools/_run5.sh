cd $GRAFT_REPO_ROOT
O=gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DIMCVT_PROF imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_prof.so 2>/dev/null
PP_VERBOSE=1 PP_LAUNCHES=8 timeout 1500 python tools/pool_probe.py 1920 1080 512 0 512:512:1:1:0 512:512:4:8:0 512:512:1000:1000:0 2>&1 | grep -v amdgpu.ids | tee $O/r03e_pool_probe.log
IMCVT_HEVC_LIB=$O/libimcvt_hevc_prof.so PP_VERBOSE=1 PP_PROF=1 PP_LAUNCHES=8 timeout 1500 python tools/pool_probe.py 1920 1080 512 0 512:512:1:1:0 512:512:4:8:0 2>&1 | grep -v amdgpu.ids | tee $O/r03e_pool_probe_prof.log
