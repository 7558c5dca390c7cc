cd $GRAFT_REPO_ROOT
O=gpurun_out
INL='__device__ __forceinline__'
v() { NAME=$1; shift; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "$@" imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_$NAME.so 2> $O/variant_$NAME.build.log || { echo "$NAME: build failed"; return; }
  IMCVT_HEVC_LIB=$O/libimcvt_hevc_$NAME.so timeout 300 python tools/trace_dump.py syn 100 70 3 0 $O/trace_$NAME.npy 2>&1 | grep -v amdgpu.ids
  IMCVT_HEVC_LIB=$O/libimcvt_hevc_$NAME.so timeout 300 python tools/gpu_parity.py 2>&1 | grep -c MISMATCH; }
v base
v inl_border "-DHDN_BORDER=$INL"
v inl_bft "-DHDN_BFT=$INL"
v inl_bts "-DHDN_BTS=$INL"
v inl_eval "-DHDN_EVAL=$INL"
v inl_both "-DHDN_BORDER=$INL" "-DHDN_EVAL=$INL"
for n in inl_border inl_bft inl_bts inl_eval inl_both; do echo "== base vs $n"; python tools/trace_diff.py $O/trace_base.npy $O/trace_$n.npy; done
