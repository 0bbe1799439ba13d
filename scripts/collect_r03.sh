#!/bin/bash
# Round-3 evidence pass on the GPU box: kernel stats + PMC (matrix pipe) of the headline and the DeepFashion bf16 step, then the
# driver's default bench line.  Outputs under gpurun_out/profiles_out/ (copied into profiles/ afterwards).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/collect_stats.sh r03_market_f32        # stand-alone launch durations (what roofline.avg_launch_us is compared with)
STATS_NOTE=scripts/two_stream_note.md bash scripts/collect_stats.sh r03g_market_f32                          # the default: encoder towers on two streams (durations include sharing)
bash scripts/collect_stats.sh r03_df256_bf16 --workload df256 --dtype bf16
DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/pmc_mfma_bench.sh r03_market_f32
bash scripts/pmc_mfma_bench.sh r03_df256_bf16 --workload df256 --dtype bf16
python bench.py > gpurun_out/profiles_out/r03_bench.json 2> gpurun_out/profiles_out/r03_bench.err
tail -c 300 gpurun_out/profiles_out/r03_bench.err
ls -la gpurun_out/profiles_out/ | tail -12
