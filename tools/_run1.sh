cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a_gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03a_gpu_tests.log
tools/gpu_variant.sh base
tools/gpu_variant.sh inl_border '-DHDN_BORDER=__device__ __forceinline__'
tools/gpu_variant.sh inl_border_nsa '-DHDN_BORDER=__device__ __forceinline__' -fno-strict-aliasing
tools/gpu_variant.sh inl_eval '-DHDN_EVAL=__device__ __forceinline__'
tools/gpu_variant.sh inl_both '-DHDN_BORDER=__device__ __forceinline__' '-DHDN_EVAL=__device__ __forceinline__'
grep -h MISMATCH gpurun_out/variant_*.log | head -20
