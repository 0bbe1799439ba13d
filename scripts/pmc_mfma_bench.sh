#!/bin/bash
# usage (on the GPU box): bash scripts/pmc_mfma_bench.sh <tag> <bench.py args...>
# one rocprofv3 PMC pass (matrix-pipe utilisation + wave-cycle split; own pass, kernel trace only) of an eager bench.py run
# -> profiles/<tag>_pmc_mfma.md (+ copy under gpurun_out/profiles_out/)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp; S=/tmp/dpig_pmc_$TAG; rm -rf $S; mkdir -p $S $R/gpurun_out/profiles_out
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $S/mfma -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-info-lines --no-graph > $S/mfma.log 2>&1
cd $R
DB="$(find $S/mfma -name '*.db' | head -1)"
if [ -z "$DB" ]; then echo "no counter database"; tail -5 $S/mfma.log; exit 1; fi
python scripts/pmc_mfma_summary.py "$DB" $TAG
sed -i "s#python bench.py --steps 2#python bench.py $* --steps 2#" profiles/${TAG}_pmc_mfma.md
cp profiles/${TAG}_pmc_mfma.md gpurun_out/profiles_out/
