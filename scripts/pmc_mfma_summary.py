"""Summarise a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES) into profiles/<tag>_pmc_mfma.md: matrix-pipe utilisation per kernel.

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the counter advances by the nominal pipe
occupancy of every MFMA issued (64 cycles for v_mfma_f32_32x32x2_f32), GRBM_GUI_ACTIVE by shader clocks per XCD."""
import collections, os, sqlite3, sys
db, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cur = sqlite3.connect(db).cursor()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for k, c, v in cur.execute("select kernel_name,counter_name,value from counters_collection"):
    agg[k][c] += v
    if c == "GRBM_GUI_ACTIVE": cnt[k] += 1
def short(n):
    n = n.replace("void ", ""); return n[:n.index("(")] if "(" in n else n
rows = []
for k, a in agg.items():
    if "dpig::" not in k or a["SQ_VALU_MFMA_BUSY_CYCLES"] == 0: continue
    gui = a["GRBM_GUI_ACTIVE"] / 8.0
    wc = max(a["SQ_WAVE_CYCLES"], 1.0)
    rows.append((a["SQ_VALU_MFMA_BUSY_CYCLES"], short(k), cnt[k], a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / gui,
                 a["SQ_ACTIVE_INST_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_WAIT_ANY"] / wc,
                 a.get("SQ_WAIT_INST_LDS", 0.0) / wc, a.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(a.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)))
rows.sort(reverse=True)
with open(os.path.join(root, "profiles", tag + "_pmc_mfma.md"), "w") as f:
    f.write("# Matrix-pipe utilisation from rocprofv3 PMC counters: %s\n\n" % tag)
    f.write("`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY "
            "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph`\n"
            "(own pass, no other trace domains).  MfmaUtil = MFMA busy cycles / (1024 SIMDs x shader cycles per XCD); the wave-cycle\n"
            "split says where a resident wave spends its time (issuing, stalled at issue = mostly behind the matrix pipe, parked on\n"
            "s_waitcnt / s_barrier).\n\n")
    f.write("| kernel | launches | MfmaUtil | wave cycles: issuing | issue-stalled | parked | (LDS-issue-stalled) | LDS conflict cycles / LDS active |\n|---|---|---|---|---|---|---|---|\n")
    for _, k, n, u, ai, wi, wa, wl, bc in rows:
        f.write("| `%s` | %d | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.2f |\n" % (k, n, 100 * u, 100 * ai, 100 * wi, 100 * wa, 100 * wl, bc))
print("ok", len(rows))
