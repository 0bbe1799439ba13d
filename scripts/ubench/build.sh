#!/bin/bash
# Build the measurement aids behind DESIGN.md section 5 (run the binaries / scripts on an MI355X):
#   mfma_peak    what the fp32 matrix pipe sustains (waves per SIMD, barrier, operand data) + s_memtime tick rate
#   hwid_probe   HW_ID / XCC_ID / LDS_ALLOC of co-resident workgroups
#   bhq32_probe  prototype of the BK = 32 halo-staged conv kernel for 128-column layers: self-check + TFLOP/s
#   libdpig_trace.so   the library with -DDPIG_TRACE: s_memtime stamps inside the conv kernels;
#                      DPIG_LIB_PATH=scripts/ubench/libdpig_trace.so python scripts/ubench/trace_run.py 256 256   (forward)
#                      DPIG_LIB_PATH=scripts/ubench/libdpig_trace.so python scripts/ubench/trace_wgrad.py 256 256 (wgrad)
set -e
cd "$(dirname "$0")"
R=../..; P=$R/disentangled-person-image-generation_amd
hipcc --offload-arch=gfx950 -O3 -w -o mfma_peak mfma_peak.hip
hipcc --offload-arch=gfx950 -O3 -w -o hwid_probe hwid_probe.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I $R/include -I $P/csrc -o bhq32_probe bhq32_probe.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDPIG_TRACE -I $R/include -I $P/csrc -o libdpig_trace.so $P/csrc/*.hip
