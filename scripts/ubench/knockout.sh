#!/bin/bash
# Knock-out experiments on the bf16 gather-GEMM (bg_kernel): rebuild the library with one resource removed at a time
# (results are WRONG by construction; only the clock is read) and time the same layers.  Run on the GPU box:
#   bash scripts/ubench/knockout.sh build    (here, cross-compiles)      bash scripts/ubench/knockout.sh run   (on the box)
R=$(cd "$(dirname "$0")/../.." && pwd); P=$R/disentangled-person-image-generation_amd
cd "$R/scripts/ubench"
if [ "$1" = build ]; then
  for ko in NONE LDS DMA BAR MFMA; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDPIG_KO_$ko -I $R/include -I $P/csrc -o libdpig_ko_$ko.so $P/csrc/*.hip &
  done; wait; ls -la libdpig_ko_*.so
else
  for ko in NONE LDS DMA BAR MFMA; do
    echo "== knock-out: $ko"; DPIG_BF16_HALO=0 DPIG_LIB_PATH=$R/scripts/ubench/libdpig_ko_$ko.so python $R/scripts/bench_conv_bf16s.py --quick 2>&1 | grep "^bf16" | cut -c1-60
  done
fi
