#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats of one bench.py workload; summary into profiles/<tag>_kernel_stats.{md,csv}
# usage: collect_stats.sh <tag> [bench.py arguments ...]        e.g.  collect_stats.sh r03a_df256_bf16 --workload df256 --dtype bf16
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
S=/tmp/dpig_prof_$TAG; rm -rf $S; mkdir -p $S $R/gpurun_out/profiles_out
timeout 600 rocprofv3 --kernel-trace --stats -d $S/stats -- python $R/bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-info-lines > $S/stats.log 2>&1
cd $R
V=$(grep -o '"value": [0-9.]*' $S/stats.log | head -1); M=$(grep -o '"ms_per_step": [0-9.]*' $S/stats.log | head -1)
python scripts/rocprof_summary.py "$(find $S/stats -name '*.db' | head -1)" $TAG "Command: \`${DPIG_TWO_STREAM:+DPIG_TWO_STREAM=$DPIG_TWO_STREAM }rocprofv3 --kernel-trace --stats -- python bench.py $* --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-info-lines\` (1 MI355X; 13 graph-replayed steps + 2 eager capture steps + init; under the profiler: $V, $M)."
if [ -n "$STATS_NOTE" ]; then cat $STATS_NOTE >> profiles/${TAG}_kernel_stats.md; fi
cp profiles/${TAG}_kernel_stats.* gpurun_out/profiles_out/
