#!/bin/bash
# Round-6 evidence pass on the GPU box (stages selected by arguments so that one gpurun call can do a bounded subset):
#   tests      pytest -m gpu (log -> gpurun_out/pytest_gpu.log)
#   traffic    HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, own passes) of the headline + the bf16 workloads -> profiles/r06_*_pmc_traffic.md,
#              profiles/roofline_traffic.json (what bench.py reports as roofline.traffic)
#   stats      rocprofv3 --kernel-trace --stats of the headline (one stream), Market bf16, DeepFashion bf16, stage-II bf16
#   pmc        matrix-pipe counters of the DeepFashion bf16 step AT HEAD with rocm-smi clock / power samples taken during the pass
#   layers     per-layer tables (Market bf16, DeepFashion bf16, Market fp32)
#   crash      scripts/diag_syncbn_graph_2rank.py (two ranks, SyncBN, captured graphs)
#   bench      the driver's command, every information line
#   guard      the guard-page sweeps: tests/guard/run_cases.py hi + lo (NaN-filled), the GPU suite per file on guard pages (both placements),
#              the two-rank eager SyncBN step with every rank's allocations guarded
#   crash50    scripts/diag_syncbn_graph_2rank.py 50 graphs -- VERDICT r5 #2's "50/50" bar for the two-rank captured step
#   dry        scripts/run_scale.sh --dry (every branch of the 8-GPU script at N = 2 over gloo on this one GPU)
# DPIG_HEAD=<git short hash> in the environment stamps the traffic entries (the GPU box has no .git)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/profiles_out
for stage in "$@"; do
case $stage in
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/pytest_gpu.log ;;
newtests)
  timeout 1200 python -m pytest tests/test_fullsize_gpu.py tests/test_golden_gpu.py tests/test_tfrecord.py "tests/test_variants_gpu.py::test_stage1_bf16_storage_mode" \
     "tests/test_variants_gpu.py::test_stage2_graph_warmup_leaves_the_learning_rates_alone" "tests/test_variants_gpu.py::test_join_side_streams_accepts_an_unindexed_device" \
     -m gpu -x -q -p no:cacheprovider --durations=15 > gpurun_out/pytest_new.log 2>&1; echo "pytest(new) rc $?"; tail -25 gpurun_out/pytest_new.log ;;
traffic)
  DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/pmc_traffic.sh r06_market_f32 market128/f32 --dtype f32
  DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/pmc_traffic.sh r06_market_bf16 market128/bf16 --workload market128 --dtype bf16
  DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/pmc_traffic.sh r06_df256_bf16 df256/bf16 --workload df256 --dtype bf16
  DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/pmc_traffic.sh r06_stage2_bf16 market128-stage2/bf16 --workload market128-stage2 --dtype bf16
  DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/pmc_traffic.sh r06_df256_wgan_gp_bf16 df256-wgan-gp/bf16 --workload df256-wgan-gp --dtype bf16
  cat profiles/roofline_traffic.json ;;
headline)
  # the round-5 headline (f32w: Winograd where it pays): traffic passes, kernel stats (one stream), matrix-pipe counters, per-layer table
  DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/pmc_traffic.sh r06_market_f32w market128/f32w --dtype f32w
  DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/pmc_traffic.sh r06_df256_f32w df256/f32w --workload df256 --dtype f32w
  DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/collect_stats.sh r06_market_f32w --dtype f32w
  DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/pmc_mfma_bench.sh r06_market_f32w --dtype f32w
  DPIG_WORKLOAD=market128 DPIG_DTYPE=f32w DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 timeout 300 python scripts/layer_table.py > gpurun_out/profiles_out/r06_layer_market_f32w.txt 2>&1
  head -4 gpurun_out/profiles_out/r06_layer_market_f32w.txt; tail -12 profiles/r06_market_f32w_pmc_mfma.md | cut -c1-160 ;;
stats)
  DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/collect_stats.sh r06_market_f32 --dtype f32
  bash scripts/collect_stats.sh r06_market_bf16 --workload market128 --dtype bf16
  bash scripts/collect_stats.sh r06_df256_bf16 --workload df256 --dtype bf16
  bash scripts/collect_stats.sh r06_stage2_bf16 --workload market128-stage2 --dtype bf16 ;;
pmc)
  # clock / power while the counter pass runs (one sample per second): MfmaUtil and achieved-of-peak can only be reconciled with the
  # clock the part actually sustained
  ( for i in $(seq 1 120); do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 1; done ) > gpurun_out/profiles_out/r06_df256_bf16_smi.jsonl &
  SMI=$!
  DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 bash scripts/pmc_mfma_bench.sh r06_df256_bf16 --workload df256 --dtype bf16
  kill $SMI 2>/dev/null
  python scripts/smi_summary.py gpurun_out/profiles_out/r06_df256_bf16_smi.jsonl >> profiles/r06_df256_bf16_pmc_mfma.md; cp profiles/r06_df256_bf16_pmc_mfma.md gpurun_out/profiles_out/ ;;
layers)
  DPIG_WORKLOAD=market128 DPIG_DTYPE=bf16 DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 timeout 300 python scripts/layer_table.py > gpurun_out/profiles_out/r06_layer_market_bf16.txt 2>&1
  DPIG_WORKLOAD=df256 DPIG_DTYPE=bf16 DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 timeout 300 python scripts/layer_table.py > gpurun_out/profiles_out/r06_layer_df256_bf16.txt 2>&1
  DPIG_WORKLOAD=market128 DPIG_DTYPE=f32w DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 timeout 300 python scripts/layer_table.py > gpurun_out/profiles_out/r06_layer_market_f32w.txt 2>&1
  DPIG_WORKLOAD=market128 DPIG_DTYPE=f32 DPIG_TWO_STREAM=0 DPIG_D_OVERLAP=0 timeout 300 python scripts/layer_table.py > gpurun_out/profiles_out/r06_layer_market_f32.txt 2>&1
  head -3 gpurun_out/profiles_out/r06_layer_*.txt ;;
guard)
  mkdir -p gpurun_out/guard
  for m in hi lo; do
    (time AMD_SERIALIZE_KERNEL=3 DPIG_GUARD_FILL=255 timeout 900 python tests/guard/run_cases.py $m) > gpurun_out/guard/cases_$m.log 2>&1
    echo "guard cases $m: OK $(grep -c '^OK' gpurun_out/guard/cases_$m.log) ERR $(grep -c '^ERR' gpurun_out/guard/cases_$m.log) $(grep '^DONE' gpurun_out/guard/cases_$m.log)"
    DPIG_GUARD=$m AMD_SERIALIZE_KERNEL=3 DPIG_GUARD_FILL=255 timeout 600 python scripts/diag_syncbn_graph_2rank.py 6 eager > gpurun_out/guard/two_rank_eager_$m.log 2>&1
    grep SUMMARY gpurun_out/guard/two_rank_eager_$m.log
    GUARD_FILE_TIMEOUT=600 bash scripts/guard_suite.sh $m > gpurun_out/guard/suite_$m.txt 2>&1; grep '^==' gpurun_out/guard/suite_$m.txt
  done ;;
crash50)
  timeout 1500 python scripts/diag_syncbn_graph_2rank.py 50 graphs > gpurun_out/crash50_graphs.log 2>&1
  grep -E "^attempt|SUMMARY" gpurun_out/crash50_graphs.log | grep -v "\[0, 0\]" | tail -12
  grep -B2 -A40 "log tail" gpurun_out/crash50_graphs.log | head -120 ;;
dry)
  bash scripts/run_scale.sh --dry 2>&1 | tee gpurun_out/scale_dry.txt | tail -60 ;;
crash)
  timeout 600 python scripts/diag_syncbn_graph_2rank.py ${CRASH_ATTEMPTS:-8} graphs > gpurun_out/crash_graphs.log 2>&1; tail -60 gpurun_out/crash_graphs.log ;;
bench)
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/profiles_out/r06_bench.json 2> gpurun_out/profiles_out/r06_bench.err
  cp gpurun_out/bench_info.jsonl gpurun_out/profiles_out/r06_bench_info.jsonl
  tail -c 300 gpurun_out/profiles_out/r06_bench.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/profiles_out/r06_bench.json") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"])
print("line bytes", len(json.dumps(d, separators=(",", ":"))))
for k, v in (d.get("info") or {}).items():
    print(" ", k, v)
PY
  ;;
esac
done
ls gpurun_out/profiles_out | tail -30
