#!/bin/bash
# Build a variant of the kernel with extra compiler flags and run the quick parity script on it.  usage: tools/gpu_variant.sh NAME flags...
NAME=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -disable-machine-licm "$@" imcvt_amd/csrc/hevc_hip.hip -o $O/libimcvt_hevc_$NAME.so 2> $O/variant_$NAME.build.log || { echo "$NAME: build failed"; tail -5 $O/variant_$NAME.build.log; exit 1; }
IMCVT_HEVC_LIB=$O/libimcvt_hevc_$NAME.so timeout 600 python tools/gpu_parity.py ${PARITY_ARGS:-} > $O/variant_$NAME.log 2>&1
echo "$NAME: rc=$? $(grep -c MISMATCH $O/variant_$NAME.log) mismatch lines; $(tail -1 $O/variant_$NAME.log)"
