"""Per kernel: mean duration and mean idle gap before the NEXT kernel starts, from a rocprofv3 --kernel-trace results .db.
   python scripts/kernel_gaps.py <results.db> [name substring ...]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table', 'view')")]
src = None
for cand in ["kernels"] + [n for n in names if "kernel" in n.lower()]:
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(%s)" % cand)]
    except Exception:
        continue
    if "start" in cols and "end" in cols and ("name" in cols or "kernel_name" in cols):
        src = (cand, "name" if "name" in cols else "kernel_name")
        break
if not src:
    print("no kernel table with start / end / name; tables:", names)
    sys.exit(1)
rows = list(cur.execute("select %s, start, end from %s order by start" % (src[1], src[0])))
pats = sys.argv[2:]
agg = {}
for i, (n, s, e) in enumerate(rows):
    if pats and not any(p in n for p in pats):
        continue
    short = n.replace("void ", "")
    short = short[:short.index("(")] if "(" in short else short[:60]
    gap = rows[i + 1][1] - e if i + 1 < len(rows) else None
    a = agg.setdefault(short, [0, 0.0, 0, 0.0, []])
    a[0] += 1; a[1] += e - s
    if gap is not None and rows[i + 1][0] == n:          # gap to the next launch of the SAME kernel (a back-to-back timing loop)
        a[2] += 1; a[3] += gap; a[4].append(gap)
durs = {}
for n, s_, e in rows:
    if pats and not any(p in n for p in pats):
        continue
    durs.setdefault(n, []).append(e - s_)
for n, d in durs.items():
    d.sort()
    q = lambda f: d[min(int(f * len(d)), len(d) - 1)] / 1e3
    print("durations [us] of %s...: n %d  min %.1f  10%% %.1f  25%% %.1f  50%% %.1f  75%% %.1f  90%% %.1f  max %.1f" % (
        n[:40], len(d), d[0] / 1e3, q(0.10), q(0.25), q(0.50), q(0.75), q(0.90), d[-1] / 1e3))
print("%-60s %6s %12s %8s %14s %12s" % ("kernel", "calls", "mean dur us", "b2b n", "mean gap us", "median gap"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    g = sorted(a[4])
    print("%-60s %6d %12.1f %8d %14.2f %12.2f" % (k[:60], a[0], a[1] / a[0] / 1e3, a[2], a[3] / max(a[2], 1) / 1e3, (g[len(g) // 2] / 1e3) if g else float("nan")))
