#!/bin/bash
# Round-4 evidence pass on the GPU box: kernel stats of the headline (one stream: stand-alone launch durations; and the default two-stream
# run), the DeepFashion bf16 step, its wgan-gp variant and the stage-II step; matrix-pipe counters of the headline; then the driver's
# bench line with every information line.  Outputs under gpurun_out/profiles_out/ (copied into profiles/ afterwards).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/collect_stats.sh r04_market_f32
STATS_NOTE=scripts/two_stream_note.md bash scripts/collect_stats.sh r04g_market_f32
bash scripts/collect_stats.sh r04_df256_bf16 --workload df256 --dtype bf16
bash scripts/collect_stats.sh r04_df256_wgan_gp_bf16 --workload df256-wgan-gp --dtype bf16
bash scripts/collect_stats.sh r04_stage2_bf16 --workload market128-stage2 --dtype bf16
bash scripts/collect_stats.sh r04_market_bf16 --workload market128 --dtype bf16
DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/pmc_mfma_bench.sh r04_market_f32
DPIG_WORKLOAD=df256 DPIG_DTYPE=bf16 DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 timeout 300 python scripts/layer_table.py > gpurun_out/profiles_out/r04_layer_df256_bf16.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/profiles_out/r04_bench.json 2> gpurun_out/profiles_out/r04_bench.err
tail -c 300 gpurun_out/profiles_out/r04_bench.err
ls -la gpurun_out/profiles_out/ | tail -20
