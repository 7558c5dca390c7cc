cd $GRAFT_REPO_ROOT
O=gpurun_out; TAG=r03s
timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_gpu_tests.log; tail -3 $O/${TAG}_gpu_tests.log
timeout 900 python bench.py > $O/${TAG}_bench_512f.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; cat $O/${TAG}_bench_512f.json | cut -c1-3000
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/rocprof_${TAG} -o ${TAG} -- python $R/bench.py --no-cpu-baseline --no-latency-view > $R/$O/${TAG}_bench_under_rocprof.json 2> $R/$O/${TAG}_rocprof.err; echo "rocprof rc=$?"
cd $R
DB=$(find $O/rocprof_${TAG} -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_trace_stats.txt && head -30 $O/${TAG}_kernel_trace_stats.txt
timeout 300 python tools/jls_bench.py 1920 1080 1 0 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_jls_bench.log
timeout 300 python tools/jls_bench.py 1920 1080 64 0 2>&1 | grep -v amdgpu.ids | tee -a $O/${TAG}_jls_bench.log
timeout 300 python tools/jls_bench.py 3840 2160 1 0 2>&1 | grep -v amdgpu.ids | tee -a $O/${TAG}_jls_bench.log
