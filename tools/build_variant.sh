#!/bin/bash
# build a variant of the library (both objects, the shipped flags + extra defines) into imcvt_amd/csrc/variants/libimcvt_hevc_<name>.so     usage: tools/build_variant.sh name [flags...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/imcvt_amd/csrc; N=$1; shift
mkdir -p $C/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -disable-machine-licm "$@" -c $C/hevc_hip.hip -o /tmp/v_$N.a.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c $C/hevc_wide.hip -o /tmp/v_$N.b.o &
wait
hipcc --offload-arch=gfx950 -fPIC -shared /tmp/v_$N.a.o /tmp/v_$N.b.o -o $C/variants/libimcvt_hevc_$N.so
rm -f /tmp/v_$N.a.o /tmp/v_$N.b.o
ls -la $C/variants/libimcvt_hevc_$N.so
