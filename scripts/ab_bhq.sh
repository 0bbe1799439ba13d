#!/bin/bash
# Same-box A/B of the 256 x 256 forward / dgrad kernels: bq<2,4> (DPIG_BF16_QH=0), bhq (1), bhq with the relaxed halo wait (2).
# Output: gpurun_out/profiles_out/r03f_bhq_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/profiles_out
O=gpurun_out/profiles_out/r03f_bhq_ab.txt
echo "(scripts/ab_bhq.sh: scripts/bench_conv_bf16q.py --quick per setting of DPIG_BF16_QH, same box, interleaved; effective TFLOP/s)" > $O
for r in 1 2; do for q in ${AB_SET:-0 1 2}; do
  echo "--- DPIG_BF16_QH=$q round $r" >> $O
  DPIG_BF16_QH=$q python scripts/bench_conv_bf16q.py --quick 2>&1 | grep -E "dec4|enc1|dec3|dec2" | sed -E 's/\| q512.*//' >> $O
done; done
cat $O
