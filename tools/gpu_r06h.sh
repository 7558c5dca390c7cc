#!/bin/bash
# round 6: the quiet lead-sink flush — whole GPU suite on the shipped library, then interleaved A/B (bench shape, one frame, 64 frames) of two single-source builds, then the region counters again
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
T=${1:-r06h}
timeout 2400 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $O/${T}_gpu_tests.log; tail -4 $O/${T}_gpu_tests.log | cut -c1-200
L=$O/${T}_quiet_ab.log; : > $L
for rep in 1 2 3; do
  for v in noquiet quiet; do
    echo "== $v" >> $L
    export IMCVT_HEVC_LIB=$R/imcvt_amd/csrc/variants/libimcvt_hevc_$v.so
    PP_LAUNCHES=2 timeout 600 python tools/pool_probe.py 1920 1080 512 0 a:a 2>&1 | grep "x 1920" >> $L
    WP_LAUNCHES=2 timeout 600 python tools/wide_probe.py 1920 1080 0 1 64 2>&1 | grep "wide 1" >> $L
  done
done
unset IMCVT_HEVC_LIB
cat $L | cut -c1-220
timeout 900 python tools/valu_dyn_mix.py --frames 64 --out $O/${T}_valu_dyn_mix.json --save-counts $O/${T}_region_counts.json > $O/${T}_dyn_mix.log 2>&1; grep -i "flushes\|share_in\|mix_weighted_cycles_simd" $O/${T}_dyn_mix.log | cut -c1-200
