#!/bin/bash
# rocprofv3 kernel trace of ONE 1080p frame and one 4K frame alone on the GPU (pipe-wave workgroups), for the record.   usage: tools/gpu_trace_1frame.sh TAG
TAG=${1:-r03zx}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
PP_LAUNCHES=2 PP_MODES=-1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/rocprof_1f_$TAG -o one -- python $R/tools/pipe_probe.py 1920 1080 0 1 > $O/${TAG}_1frame_under_rocprof.log 2>&1; echo "rc=$?"
cd $R
DB=$(find $O/rocprof_1f_$TAG -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/${TAG}_kernel_trace_1frame.txt && head -20 $O/${TAG}_kernel_trace_1frame.txt
grep -v amdgpu.ids $O/${TAG}_1frame_under_rocprof.log | tail -2
rm -rf $O/rocprof_1f_$TAG
