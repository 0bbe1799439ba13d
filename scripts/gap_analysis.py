"""Idle time between kernels inside a replayed step, from a rocprofv3 kernel trace in CSV form.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
    python scripts/gap_analysis.py <dir> [out.md]

A step is delimited by `dpig::adam_kernel` (one per optimizer op, two per step).  For the last few steps the script
reports wall span, the union of kernel intervals (busy), their difference (idle), the gap histogram and the kernels
that precede the largest share of idle time."""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
paths = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not paths:
    sys.exit("no *kernel_trace.csv under " + d)
rows = []
with open(paths[0], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
nsteps = min(6, len(adam) // 2 - 1)
lines = []
tot_wall = tot_busy = 0
gaps = []
after = defaultdict(lambda: [0, 0])
for s in range(nsteps):
    lo = adam[-1 - 2 * (s + 1)] + 1        # first kernel after the closing adam of the previous step
    hi = adam[-1 - 2 * s]                  # closing adam of this step
    seg = rows[lo:hi + 1]
    t0 = rows[lo - 1][1]                   # end of the previous step's closing adam: the host-side replay gap counts
    wall = seg[-1][1] - t0
    busy, end = 0, t0
    for i, (a, b, n) in enumerate(seg):
        if a > end:
            g = a - end
            gaps.append(g)
            p = seg[i - 1][2] if i else rows[lo - 1][2] + " [previous step]"
            p = p[:p.index("(")] if "(" in p else p
            after[p][0] += g
            after[p][1] += 1
            busy += b - a
        else:
            busy += max(0, b - max(a, end))
        end = max(end, b)
    tot_wall += wall
    tot_busy += busy
    lines.append("| %d | %d | %.3f | %.3f | %.3f |" % (s, len(seg), wall / 1e6, busy / 1e6, (wall - busy) / 1e6))
hdr = ["# idle time between kernels of a replayed step", "",
       "| step (from the end) | kernels | wall ms | busy ms | idle ms |", "|---|---|---|---|---|"] + lines
idle = tot_wall - tot_busy
hdr += ["", "mean idle per step %.3f ms = %.2f %% of the step; %d gaps per step, median %.2f us, p90 %.2f us" % (
    idle / nsteps / 1e6, 100.0 * idle / tot_wall, len(gaps) // nsteps, sorted(gaps)[len(gaps) // 2] / 1e3,
    sorted(gaps)[int(len(gaps) * 0.9)] / 1e3), "", "| idle after kernel | us per step | gaps per step | avg us |", "|---|---|---|---|"]
for k, v in sorted(after.items(), key=lambda kv: -kv[1][0])[:12]:
    hdr.append("| `%s` | %.1f | %d | %.2f |" % (k.replace("void ", "")[:80], v[0] / nsteps / 1e3, v[1] // nsteps, v[0] / v[1] / 1e3))
text = "\n".join(hdr) + "\n"
print(text)
if out:
    open(out, "w").write(text)
