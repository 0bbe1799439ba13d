# usage (on the GPU box): bash scripts/collect_profile_r02.sh <tag> <bench.py args...>
# rocprofv3 --kernel-trace --stats of one bench.py command -> profiles/<tag>_kernel_stats.{md,csv} (+ copy under gpurun_out/)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; S=/tmp/dpig_prof_$TAG; rm -rf $S; mkdir -p $S
timeout 900 rocprofv3 --kernel-trace --stats -d $S/stats -- python $R/bench.py "$@" > $S/stats.log 2>&1
cd $R
python scripts/rocprof_summary.py "$(find $S/stats -name '*.db' | head -1)" $TAG "Command: \`rocprofv3 --kernel-trace --stats -- python bench.py $*\` (1 MI355X; under the profiler: $(grep -o '"value": [0-9.]*' $S/stats.log | head -1), $(grep -o '"ms_per_step": [0-9.]*' $S/stats.log | head -1))."
mkdir -p gpurun_out/profiles_out; cp profiles/${TAG}_* gpurun_out/profiles_out/; tail -c 600 $S/stats.log > gpurun_out/profiles_out/${TAG}_bench_tail.log
