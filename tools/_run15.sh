cd $GRAFT_REPO_ROOT
O=gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DWD_TICKS=50000000ull imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_wd1s.so 2>&1 | grep -i error
for i in 1 2 3; do
IMCVT_HEVC_LIB=$O/libimcvt_hevc_wd1s.so PP_LAUNCHES=12 timeout 300 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep -v amdgpu.ids | grep "watchdog:\|when it gave\|clocks\|^512" | cut -c1-300 | tee -a $O/r03o_wd_probe.log
done
