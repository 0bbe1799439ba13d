cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; S=/tmp/dpig_prof_bf; rm -rf $S; mkdir -p $S
timeout 600 rocprofv3 --kernel-trace --stats -d $S/stats -- python $R/bench.py --dtype bf16 --steps 20 --warmup 3 > $S/stats.log 2>&1
cd $R
python scripts/rocprof_summary.py "$(find $S/stats -name '*.db' | head -1)" r01_bf16 "Command: \`rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 --steps 20 --warmup 3\` (information line: conv GEMMs on the bf16 matrix pipe, fp32 tensors; Market 128x64 bs=16, 1 MI355X; under the profiler: $(grep -o '"value": [0-9.]*' $S/stats.log | head -1), $(grep -o '"ms_per_step": [0-9.]*' $S/stats.log | head -1))."
mkdir -p gpurun_out/profiles_out; cp profiles/r01_bf16_* gpurun_out/profiles_out/
