"""Summarise a rocprofv3 results .db (kernel trace) into profiles/<tag>_kernel_stats.csv + .md."""
import csv, os, sqlite3, sys
db_path, tag = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cur = sqlite3.connect(db_path).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc"))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
with open(os.path.join(root, "profiles", tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows: w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.4f" % r[4]])
def short(n):
    n = n.replace("void ", "")
    return (n[:n.index("(")] if "(" in n and not n.startswith("at::") else n[:70])[:90]
with open(os.path.join(root, "profiles", tag + "_kernel_stats.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary: %s\n\n%s\n\n" % (tag, note))
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:25]:
        f.write("| `%s` | %d | %.2f | %.1f | %.2f |\n" % (short(r[0]), r[1], r[2] / 1e3, r[3], r[4]))
    tot = sum(r[2] for r in rows)
    dp = sum(r[2] for r in rows if "dpig::" in r[0])
    f.write("\nTotal GPU kernel time %.1f ms; hand-written dpig:: kernels %.1f %% of it.\n" % (tot / 1e3, 100.0 * dp / tot))
print("ok")
