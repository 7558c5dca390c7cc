#!/bin/bash
# round-6 measurement pass on ONE box: GPU tests; counters on the bench workload and shape (-> profiles/pmc_*.json, which bench.py reads); the dynamic opcode mix from a
# region-counter build (-> profiles/valu_dyn_mix.json, the issue peak of the bench line); the bench line; the same command under rocprofv3 --kernel-trace --stats; JPEG-LS;
# the scale prediction (per-GPU shares through the RCCL path); a weak-scaling line (one GPU's worth); the A/B probes of the round.   usage: gpurun -- bash tools/gpu_r06fin.sh TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=gpurun_out; mkdir -p $R/$O; cd $R
T=${1:-r06fin}
timeout 3000 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/${T}_gpu_tests.log; tail -3 $O/${T}_gpu_tests.log | cut -c1-200
bash tools/gpu_pmc.sh ${T} 1920 1080 512 0 > $O/${T}_pmc.log 2>&1; tail -3 $O/${T}_pmc.log | cut -c1-300
python tools/pmc_issue.py $O/${T}_pmc_sq.txt 512 1920 1080 0 "the bench's launch shape: 512 main + 512 helper workgroups" > $O/${T}_pmc_issue.json; cat $O/${T}_pmc_issue.json | cut -c1-300
cp $O/${T}_pmc_issue.json profiles/pmc_issue.json; cp $O/${T}_pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python tools/valu_dyn_mix.py --frames 64 --out $O/${T}_valu_dyn_mix.json --save-counts $O/${T}_region_counts.json > $O/${T}_valu_dyn_mix.log 2>&1; grep -i "flushes\|share_in\|mix_weighted_cycles_simd" $O/${T}_valu_dyn_mix.log | cut -c1-200
[ -s $O/${T}_valu_dyn_mix.json ] && cp $O/${T}_valu_dyn_mix.json profiles/valu_dyn_mix.json
timeout 1500 python bench.py > $O/${T}_bench_512f.json 2> $O/${T}_bench.err; echo "bench rc=$?"; cut -c1-3000 $O/${T}_bench_512f.json
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/rocprof_${T} -o ${T} -- python $R/bench.py --no-cpu-baseline --no-latency-view > $R/$O/${T}_bench_under_rocprof.json 2> $R/$O/${T}_rocprof.err; echo "rocprof rc=$?"
cd $R
DB=$(find $O/rocprof_${T} -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${T}_kernel_trace_stats.txt && head -20 $O/${T}_kernel_trace_stats.txt | cut -c1-200
rm -rf $O/rocprof_${T} $O/pmc_${T}_*/
( timeout 300 python tools/jls_bench.py 1920 1080 1 0; timeout 300 python tools/jls_bench.py 1920 1080 64 0; timeout 300 python tools/jls_bench.py 3840 2160 1 0 ) 2>&1 | grep -v amdgpu.ids | tee $O/${T}_jls_bench.log
timeout 1500 python tools/scale_predict.py --out $O/${T}_scale_prediction.json 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-1500
timeout 900 python bench.py --scaling weak --frames 512 --no-cpu-baseline --no-latency-view > $O/${T}_bench_weak_512f_per_gpu.json 2> $O/${T}_bench_weak.err; echo "weak rc=$?"; cut -c1-600 $O/${T}_bench_weak_512f_per_gpu.json
timeout 1200 python tools/r06_ab.py wide split partners --reps 2 > $O/${T}_ab.log 2>&1; echo "ab rc=$?" >> $O/${T}_ab.log; cat $O/${T}_ab.log | cut -c1-800
