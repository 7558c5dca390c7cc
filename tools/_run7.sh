cd $GRAFT_REPO_ROOT
O=gpurun_out
PP_VERBOSE=1 PP_LAUNCHES=14 timeout 1500 python tools/pool_probe.py 1920 1080 512 0 512:512:2:4:0 2>&1 | grep -v amdgpu.ids | tee $O/r03f_pool_probe.log
bash tools/_run6.sh 2>&1 | tee $O/r03f_variants.log
