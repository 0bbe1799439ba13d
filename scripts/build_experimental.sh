#!/bin/bash
# The library with the experimental kernels compiled in (today: bhq32_kernel, -DDPIG_EXPERIMENTAL_BHQ32) -> scripts/ubench/libdpig_exp.so;
# the shipped libdpig_hip.so never contains them.  Use with DPIG_LIB_PATH=scripts/ubench/libdpig_exp.so (see scripts/check_bhq32.py).
set -e
cd "$(dirname "$0")/.."
P=disentangled-person-image-generation_amd
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDPIG_EXPERIMENTAL_BHQ32 -I include -I $P/csrc -o scripts/ubench/libdpig_exp.so $P/csrc/*.hip
ls -la scripts/ubench/libdpig_exp.so
