# usage (GPU box): bash scripts/pmc_l2.sh <tag>: L2 request / hit counters of the bf16 conv micro-benchmark (own PMC pass)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; S=/tmp/dpig_l2_$TAG; rm -rf $S; mkdir -p $S
timeout 600 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -d $S/l2 -- python $R/scripts/bench_conv_bf16s.py --quick > $S/l2.log 2>&1
cd $R
python - "$(find $S/l2 -name '*.db' | head -1)" <<'PY' > gpurun_out/${TAG}_l2.txt 2>&1
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for k, c, v in cur.execute("select kernel_name,counter_name,value from counters_collection"):
    agg[k][c] += v
    if c == "GRBM_GUI_ACTIVE": cnt[k] += 1
for k, a in agg.items():
    if "bfk" not in k: continue
    n = max(cnt[k], 1)
    print(k[:60], "launches", n, {c: "%.3e" % (v / n) for c, v in a.items()})
PY
tail -3 $S/l2.log >> gpurun_out/${TAG}_l2.txt
