#!/bin/bash
# Knock-out experiments on the large-tile bf16 gather-GEMM (bq_kernel): the library rebuilt with one resource of the k-loop
# removed at a time (results WRONG by construction; only the clock is read), same layers, same data.
#   bash scripts/ubench/knockout_q.sh build   (here: cross-compiles)     bash scripts/ubench/knockout_q.sh run   (GPU box)
R=$(cd "$(dirname "$0")/../.." && pwd); P=$R/disentangled-person-image-generation_amd
cd "$R/scripts/ubench"
KOS="${KOS:-NONE WAIT DMA LDS BAR MFMA EPI PRIO STORE EPILDS}"
if [ "$1" = build ]; then
  for ko in $KOS; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDPIG_QKO_$ko -I $R/include -I $P/csrc -o libdpig_qko_$ko.so $P/csrc/*.hip &
  done; wait; ls -la libdpig_qko_*.so
else
  for ko in $KOS; do
    echo "== knock-out: $ko"; DPIG_LIB_PATH=$R/scripts/ubench/libdpig_qko_$ko.so python $R/scripts/bench_conv_bf16q.py --quick 2>&1 | grep "|" | cut -c1-175
  done
fi
