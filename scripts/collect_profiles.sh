#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + the three PMC passes of bench.py, summaries into profiles/
# and a copy under gpurun_out/profiles_out/ (the .db files stay on the box).   usage: collect_profiles.sh <tag>
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
S=/tmp/dpig_prof; rm -rf $S; mkdir -p $S $R/gpurun_out/profiles_out
timeout 600 rocprofv3 --kernel-trace --stats -d $S/stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-info-lines > $S/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $S/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-info-lines --no-graph > $S/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $S/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-info-lines --no-graph > $S/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $S/mfma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-info-lines --no-graph > $S/mfma.log 2>&1
cd $R
V=$(grep -o '"value": [0-9.]*' $S/stats.log | head -1); M=$(grep -o '"ms_per_step": [0-9.]*' $S/stats.log | head -1)
python scripts/rocprof_summary.py "$(find $S/stats -name '*.db' | head -1)" $TAG "Command: \`rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-info-lines\` (hipGraph replay: 23 graph-replayed G+D steps + 2 eager capture warm-up steps + init; Market 128x64 bs=16 fp32, 1 MI355X). Same command as the BENCH line of this round minus the CPU-baseline / roofline legs; under the profiler: $V, $M."
python scripts/pmc_summary.py "$(find $S/fetch -name '*.db' | head -1)" "$(find $S/write -name '*.db' | head -1)" $TAG
python scripts/pmc_mfma_summary.py "$(find $S/mfma -name '*.db' | head -1)" $TAG
cp profiles/${TAG}_* profiles/roofline_traffic.json gpurun_out/profiles_out/
