"""usage: python scripts/kernel_names.py <rocprofv3 .db>  -- launches and mean duration per kernel name (quick look at what a micro-benchmark ran)."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = "select s.kernel_name, count(*), avg(d.end - d.start) / 1000.0 from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 2 desc" % (kd, ks)
for name, n, us in cur.execute(q):
    print("%6d x %8.1f us  %s" % (n, us, name[:110]))
