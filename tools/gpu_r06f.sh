#!/bin/bash
# round 6, sixth pass: which stream a launch runs on (null stream / a stream of the caller's / the host-pointer path's), with and without the per-stream pre-warm
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06f}
L=$O/${T}_streams.log; : > $L
for K in 1 0 1 0; do
  echo "== IMCVT_HEVC_STREAM_PREWARM=$K" >> $L
  IMCVT_HEVC_STREAM_PREWARM=$K timeout 900 python tools/r06_ab.py streams --reps 3 2>&1 | grep probe >> $L
done
cat $L | cut -c1-600
