cd $GRAFT_REPO_ROOT
O=gpurun_out; TAG=r03s
bash tools/gpu_pmc.sh ${TAG} 1920 1080 512 0 2>&1 | tail -60
python tools/pmc_issue.py $O/${TAG}_pmc_sq.txt 512 1920 1080 0 "the bench's launch shape: 512 main + 448 helper workgroups" > $O/${TAG}_pmc_issue.json; cat $O/${TAG}_pmc_issue.json
timeout 1500 python tools/scale_predict.py --out $O/${TAG}_scale_prediction.json 2>&1 | grep -v amdgpu.ids | tail -8
