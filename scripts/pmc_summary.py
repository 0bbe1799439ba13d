"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs into profiles/ (per-kernel HBM traffic per launch).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-byte requests at 64 bytes for wide
coalesced reads -> doubled; WRITE_SIZE is used as reported (uncalibrated).  Units: the counters are in KB."""
import json, os, sqlite3, sys
fetch_db, write_db, tag = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def per_kernel(db, name):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for k, n, s in cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (name,)):
        out[k] = (n, s)
    return out
f = per_kernel(fetch_db, "FETCH_SIZE"); w = per_kernel(write_db, "WRITE_SIZE")
rows = []
for k in f:
    if "dpig::" not in k: continue
    n, fs = f[k]; wn, ws = w.get(k, (0, 0.0))
    rd = 2.0 * fs * 1024.0 / n; wr = (ws * 1024.0 / wn) if wn else 0.0
    rows.append((k, n, fs * 1024.0 / n, rd, wr, rd + wr))
rows.sort(key=lambda r: -r[1] * r[5])
def short(n):
    n = n.replace("void ", ""); return n[:n.index("(")] if "(" in n else n
with open(os.path.join(root, "profiles", tag + "_pmc_traffic.md"), "w") as fh:
    fh.write("# HBM traffic per launch from rocprofv3 PMC counters: %s\n\n" % tag)
    fh.write("Two separate passes (`--pmc FETCH_SIZE --kernel-trace`, `--pmc WRITE_SIZE --kernel-trace`) of\n"
             "`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph` (5 eager steps incl. graph-free warm-up + init).\n"
             "read = 2 x FETCH_SIZE (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md), write = WRITE_SIZE (uncalibrated).\n\n")
    fh.write("| kernel | launches | raw FETCH KB/launch | read MB/launch (corrected) | write MB/launch | total MB/launch |\n|---|---|---|---|---|---|\n")
    for k, n, raw, rd, wr, tot in rows[:16]:
        fh.write("| `%s` | %d | %.0f | %.2f | %.2f | %.2f |\n" % (short(k), n, raw / 1024.0, rd / 1e6, wr / 1e6, tot / 1e6))
js = {}
for k, n, raw, rd, wr, tot in rows:
    if "gather_gemm_kernel<false, true, false, 0>" in k: js["conv_fwd_hbm_bytes_per_launch"] = round(tot)
    if "gather_gemm_kernel<true, true, false, 0>" in k: js["conv_dgrad_hbm_bytes_per_launch"] = round(tot)
    if "wgrad_kernel<true, false, false, true, 0>" in k: js["conv_wgrad_hbm_bytes_per_launch"] = round(tot)
js["source"] = "profiles/%s_pmc_traffic.md" % tag
if not os.environ.get("DPIG_KEEP_TRAFFIC_JSON") and "conv_fwd_hbm_bytes_per_launch" in js:      # (only the headline workload owns the file)
    json.dump(js, open(os.path.join(root, "profiles", "roofline_traffic.json"), "w"), indent=1)
print(js)
