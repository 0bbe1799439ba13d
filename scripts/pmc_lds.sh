# usage (GPU box): bash scripts/pmc_lds.sh <tag>: LDS activity counters of the bf16 conv micro-benchmark (own PMC pass)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; S=/tmp/dpig_lds_$TAG; rm -rf $S; mkdir -p $S
env "$@" timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE -d $S/l -- python $R/scripts/bench_conv_bf16s.py --quick > $S/l.log 2>&1
cd $R
python - "$(find $S/l -name '*.db' | head -1)" <<'PY' > gpurun_out/${TAG}_lds.txt 2>&1
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for k, c, v in cur.execute("select kernel_name,counter_name,value from counters_collection"):
    agg[k][c] += v
    if c == "GRBM_GUI_ACTIVE": cnt[k] += 1
for k, a in agg.items():
    if "bfk" not in k or "shadow" in k or "sum" in k: continue
    gui = a["GRBM_GUI_ACTIVE"] / 8.0          # shader cycles per XCD, summed over launches
    print(k[:50], "launches", cnt[k])
    print("   LdsUtil (IDX_ACTIVE / (cycles x 256 CUs)) = %.1f %%" % (100 * a["SQ_LDS_IDX_ACTIVE"] / (gui * 256)))
    print("   data fifo full %.1f %%  cmd fifo full %.1f %%  (of cycles x CUs)" % (100 * a["SQ_LDS_DATA_FIFO_FULL"] / (gui * 256), 100 * a["SQ_LDS_CMD_FIFO_FULL"] / (gui * 256)))
    print("   LDS instructions %.3e, bank conflict cycles %.3e, addr conflict %.3e" % (a["SQ_INSTS_LDS"], a["SQ_LDS_BANK_CONFLICT"], a["SQ_LDS_ADDR_CONFLICT"]))
PY
tail -2 $S/l.log >> gpurun_out/${TAG}_lds.txt
