# usage (GPU box): bash scripts/pmc_wino4.sh <tag> -- two PMC passes (matrix pipe / waits; LDS) over the F(2x2) vs F(4x4) layer
# microbenchmark (kernel trace only) -> gpurun_out/<tag>_pmc_wino4.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; S=/tmp/dpig_pmc4_$TAG; rm -rf $S; mkdir -p $S
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $S/a -- python $R/scripts/bench_conv_wino4.py > $S/a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE -d $S/b -- python $R/scripts/bench_conv_wino4.py > $S/b.log 2>&1
cd $R
python - "$(find $S/a -name '*.db' | head -1)" "$(find $S/b -name '*.db' | head -1)" <<'PY' > gpurun_out/${TAG}_pmc_wino4.txt 2>&1
import sqlite3, sys, collections
def load(path):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    q = None
    for t in tabs:
        if t.startswith("rocpd_pmc_event") or t == "counters_collection":
            q = t
    try:
        rows = cur.execute("select kernel_name,counter_name,value from counters_collection")
    except Exception:
        rows = cur.execute("select k.kernel_name, c.name, e.value from rocpd_pmc_event e join rocpd_info_pmc c on e.pmc_id=c.id join rocpd_kernel_dispatch d on e.event_id=d.event_id join rocpd_info_kernel_symbol k on d.kernel_id=k.id")
    for k, c, v in rows:
        agg[k][c] += v
        if c == "GRBM_GUI_ACTIVE": cnt[k] += 1
    return agg, cnt
A, ca = load(sys.argv[1]); B, cb = load(sys.argv[2])
for k in A:
    if "wino" not in k or "filter" in k or "reduce" in k: continue
    a, b = A[k], B.get(k, {})
    gui = a["GRBM_GUI_ACTIVE"] / 8.0
    print(k[:70], "launches", ca[k])
    print("   MfmaUtil %.1f %%   VALU active (per SIMD) %.1f %%   VALU insts/wave-cycle %.4f" % (100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256 * 4), 100 * a["SQ_ACTIVE_INST_VALU"] / (gui * 256 * 4) , a["SQ_INSTS_VALU"] / max(a["SQ_WAVE_CYCLES"], 1)))
    print("   wave cycles: waiting on any inst %.1f %%, waiting on LDS %.1f %%, issuing %.1f %%" % (100 * a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], 100 * a["SQ_WAIT_INST_LDS"] / a["SQ_WAVE_CYCLES"], 100 * a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"]))
    if b:
        g2 = b["GRBM_GUI_ACTIVE"] / 8.0
        print("   LdsUtil (IDX_ACTIVE / (cycles x 256 CUs)) %.1f %%  bank conflict %.1f %% of cycles  data fifo full %.1f %%  cmd fifo full %.1f %%  LDS inst active %.1f %%" % (
            100 * b["SQ_LDS_IDX_ACTIVE"] / (g2 * 256), 100 * b["SQ_LDS_BANK_CONFLICT"] / (g2 * 256), 100 * b["SQ_LDS_DATA_FIFO_FULL"] / (g2 * 256), 100 * b["SQ_LDS_CMD_FIFO_FULL"] / (g2 * 256), 100 * b["SQ_ACTIVE_INST_LDS"] / (g2 * 256 * 4)))
PY
tail -3 $S/a.log >> gpurun_out/${TAG}_pmc_wino4.txt
cat gpurun_out/${TAG}_pmc_wino4.txt
