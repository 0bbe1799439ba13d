#!/bin/bash
# Knock-out timing of the F(3x3,2x2) filter-gradient kernel's staging arithmetic (dev aid; run on the GPU box: gpurun -- bash scripts/ko_wgrad.sh).
# Rebuilds dpig_conv_wino.o with -DDPIG_WINO4_KNOCKOUT into a scratch copy of the library and runs scripts/bench_conv_wino.py per variant.
set -e
cd "$(dirname "$0")/.."
PKG=disentangled-person-image-generation_amd
cp $PKG/libdpig_hip.so /tmp/libdpig_hip.so.keep
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -I $PKG/csrc -DDPIG_WINO4_KNOCKOUT -c $PKG/csrc/dpig_conv_wino.hip -o /tmp/wino_ko.o
objs=$(ls build/obj/*.o | grep -v "dpig_conv_wino.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/libdpig_hip.so $objs /tmp/wino_ko.o
for ko in ${KOS:-0 1 2 3 7}; do
    echo "== DPIG_WINO_WG_KO=$ko  (1 no x arithmetic, 2 no dy arithmetic, 4 no ds_bpermute)"
    DPIG_WINO_WG_KO=$ko python scripts/bench_conv_wino.py 2>&1 | grep "128x64 C128\|dec3\|dec4\|sum of" | cut -c1-30,96-140,180-260
done
cp /tmp/libdpig_hip.so.keep $PKG/libdpig_hip.so
