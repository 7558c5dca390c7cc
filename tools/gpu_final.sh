#!/bin/bash
# Final measurement pass of a round on one box: counters on the bench workload (-> profiles/pmc_*.json, which bench.py reads), the bench line,
# the same command under rocprofv3 --kernel-trace --stats, JPEG-LS timings, scale prediction.   usage: tools/gpu_final.sh TAG   (on the GPU box: gpurun -- tools/gpu_final.sh r03z)
cd $GRAFT_REPO_ROOT
O=gpurun_out; TAG=${1:-r04z}
bash tools/gpu_pmc.sh ${TAG} 1920 1080 512 0 > $O/${TAG}_pmc.log 2>&1; tail -3 $O/${TAG}_pmc.log
python tools/pmc_issue.py $O/${TAG}_pmc_sq.txt 512 1920 1080 0 "the bench's launch shape: 512 main + 448 helper workgroups" > $O/${TAG}_pmc_issue.json; cat $O/${TAG}_pmc_issue.json
cp $O/${TAG}_pmc_issue.json profiles/pmc_issue.json; cp $O/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
timeout 1200 python bench.py > $O/${TAG}_bench_512f.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-3500 $O/${TAG}_bench_512f.json
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/rocprof_${TAG} -o ${TAG} -- python $R/bench.py --no-cpu-baseline --no-latency-view > $R/$O/${TAG}_bench_under_rocprof.json 2> $R/$O/${TAG}_rocprof.err; echo "rocprof rc=$?"
cd $R
DB=$(find $O/rocprof_${TAG} -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_trace_stats.txt && head -30 $O/${TAG}_kernel_trace_stats.txt
rm -rf $O/rocprof_${TAG} $O/pmc_${TAG}_*/
( timeout 300 python tools/jls_bench.py 1920 1080 1 0; timeout 300 python tools/jls_bench.py 1920 1080 64 0; timeout 300 python tools/jls_bench.py 3840 2160 1 0 ) 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_jls_bench.log
timeout 1500 python tools/scale_predict.py --out $O/${TAG}_scale_prediction.json 2>&1 | grep -v amdgpu.ids | tail -3
