cd $GRAFT_REPO_ROOT
O=gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DIMCVT_NO_PRIO imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_noprio.so 2>&1 | grep -i error
IMCVT_HEVC_LIB=$O/libimcvt_hevc_noprio.so PP_VERBOSE=1 PP_LAUNCHES=16 timeout 400 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids | grep "^512\|slowest" | cut -c1-420 | tee $O/r03p_noprio_probe.log
IMCVT_HEVC_LIB=$O/libimcvt_hevc_noprio.so PP_LAUNCHES=2 timeout 400 python tools/pool_probe.py 1920 1080 1000 0 a:a 2>&1 | grep -v amdgpu.ids | grep "^1000" | cut -c1-300 | tee -a $O/r03p_noprio_probe.log
