#!/bin/bash
# usage (GPU box): bash scripts/pmc_traffic.sh <tag> <workload>/<dtype> [bench.py args...] -- the two HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE; own
# passes, kernel trace only -- never combined with other trace domains) of an eager bench.py run -> profiles/<tag>_pmc_traffic.md and the
# <workload>/<dtype> entry of profiles/roofline_traffic.json (what bench.py reports as roofline.traffic for that workload).
TAG=$1; KEY=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp; S=/tmp/dpig_traffic_$TAG; rm -rf $S; mkdir -p $S $R/gpurun_out/profiles_out
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $S/fetch -- python $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-info-lines --no-graph > $S/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $S/write -- python $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-info-lines --no-graph > $S/write.log 2>&1
cd $R
DPIG_TRAFFIC_KEY=$KEY python scripts/pmc_summary.py "$(find $S/fetch -name '*.db' | head -1)" "$(find $S/write -name '*.db' | head -1)" $TAG
sed -i "s#python bench.py --steps 2 --warmup 1#python bench.py $* --steps 1 --warmup 1#" profiles/${TAG}_pmc_traffic.md
cp profiles/${TAG}_pmc_traffic.md profiles/roofline_traffic.json gpurun_out/profiles_out/
