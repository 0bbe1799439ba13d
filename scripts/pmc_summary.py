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
# profiles/roofline_traffic.json: {"entries": {"<workload>/<dtype>": {hbm_bytes_per_launch of the dominant kernel class, ...}}} -- what
# bench.py reports as roofline.traffic for that workload (DPIG_TRAFFIC_KEY names the entry; the dtype picks the kernel family)
key = os.environ.get("DPIG_TRAFFIC_KEY", "market128/f32")
dtype = key.split("/")[1]
FAMILY = {"f32": (["gather_gemm_kernel<false, true, false, 0>"], ["conv_fwd_mfma"]),
          "f32w": (["wino4_kernel"], ["conv_fwd_wino4", "conv_dgrad_wino4"]),      # (the dominant class since the F(4x4,3x3) kernel: bench.py picks it by time)
          "bf16": (["bhq_kernel", "bhq32_kernel", "bq_kernel", "bh_kernel", "bg8_kernel", "bg8d_kernel", "bg_kernel", "bg8_multi_kernel", "bg8d_multi_kernel",
                    "bg_multi_kernel"],
                   ["conv_fwd_bf16", "conv_dgrad_bf16"])}
pats, classes = FAMILY.get(dtype, FAMILY["f32"])
def base(n):                      # "dpig::bfk::bq_kernel<2, 4>" -> "bq_kernel"
    n = short(n)
    n = n[:n.index("<")] if "<" in n else n
    return n.split("::")[-1]
sel = [r for r in rows if (base(r[0]) in pats if dtype != "f32" else any(p_ in r[0] for p_ in pats))]
sel = [r for r in sel if "wino_filter" not in r[0] and "wino_wgrad" not in r[0]]
path = os.path.join(root, "profiles", "roofline_traffic.json")
try:
    js = json.load(open(path))
except Exception:
    js = {}
if "entries" not in js:
    js = {"entries": {}}
if sel:
    nl = sum(r[1] for r in sel)
    head = os.environ.get("DPIG_HEAD") or os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % root).read().strip() or "?"
    # the kernel sources this figure was measured on: bench.py reports the entry only while csrc/ hashes to the same value
    sys.path.insert(0, root)
    import bench as _bench
    ent = {"hbm_bytes_per_launch": round(sum(r[1] * r[5] for r in sel) / nl), "launches": nl, "classes": classes,
           "kernels": sorted(set(short(r[0]) for r in sel)), "source": "profiles/%s_pmc_traffic.md" % tag, "head": head,
           "csrc_sha": _bench.csrc_sha()}
    if dtype == "f32w":           # the Winograd filter gradient beside the forward / dgrad class
        wg = [r for r in rows if base(r[0]) == "wino_wgrad_kernel"]
        if wg:
            nw = sum(r[1] for r in wg)
            ent["wgrad"] = {"hbm_bytes_per_launch": round(sum(r[1] * r[5] for r in wg) / nw), "launches": nw, "classes": ["conv_wgrad_wino"],
                            "kernels": ["dpig::wino::wino_wgrad_kernel"]}
    js["entries"][key] = ent
    json.dump(js, open(path, "w"), indent=1)
print(key, js["entries"].get(key))
