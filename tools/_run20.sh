#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; TAG=r03x
bash tools/gpu_pmc.sh ${TAG} 1920 1080 512 0 2>&1 | tail -70
python tools/pmc_issue.py $O/${TAG}_pmc_sq.txt 512 1920 1080 0 "the bench's launch shape: 512 main + 448 helper workgroups" > $O/${TAG}_pmc_issue.json; cat $O/${TAG}_pmc_issue.json
