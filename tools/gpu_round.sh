#!/bin/bash
# One GPU-box pass: parity tests, bench line, rocprofv3 kernel-trace summary, phase cycles.  usage: tools/gpu_round.sh TAG [frames]
TAG=${1:-rXX}; FR=${2:-1000}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_gpu_tests.log
tail -3 $O/${TAG}_gpu_tests.log
timeout 900 python bench.py --frames $FR > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; cat $O/${TAG}_bench.json
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/rocprof_${TAG} -o ${TAG} -- python $R/bench.py --frames $FR --no-cpu-baseline > $O/${TAG}_rocprof_bench.json 2> $O/${TAG}_rocprof.err; echo "rocprof rc=$?"
cd $R
DB=$(find $O/rocprof_${TAG} -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_trace_stats.txt && cat $O/${TAG}_kernel_trace_stats.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm -DIMCVT_PROF imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_prof.so 2>/dev/null
( IMCVT_HEVC_LIB=$O/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1 0;  IMCVT_HEVC_LIB=$O/libimcvt_hevc_prof.so timeout 300 python tools/prof_phases.py 512 256 1024 0 ) > $O/${TAG}_phase_cycles.log 2>&1
cat $O/${TAG}_phase_cycles.log
