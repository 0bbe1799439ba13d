#!/bin/bash
# Same-box A/B of the two-stream encoder towers (DPIG_TWO_STREAM=0/1): Market stage-I fp32 (the headline) and bf16, stage-II bf16.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
O=gpurun_out/ab_two_stream.txt; : > $O
run() { echo "== ${AB_VAR:-DPIG_TWO_STREAM}=$1 $2" >> $O; env ${AB_VAR:-DPIG_TWO_STREAM}=$1 timeout 200 python bench.py $2 --no-cpu-baseline --no-roofline --no-info-lines 2>&1 | grep -o '"value": [0-9.]*, \|"ms_per_step": [0-9.]*\|Error.*\|error.*' | tr '\n' ' ' >> $O; echo >> $O; }
for r in 1 2; do
  for w in ${AB_W:-f32 bf16 stage2}; do
    case $w in
      f32) a="--steps 20 --warmup 5";;
      bf16) a="--dtype bf16 --steps 30 --warmup 5";;
      stage2) a="--workload market128-stage2 --dtype bf16 --steps 10 --warmup 3";;
      sampling) a="--workload market128-sampling --dtype bf16 --steps 20 --warmup 3";;
    esac
    for t in 0 1; do run $t "$a"; done
  done
done
cat $O
