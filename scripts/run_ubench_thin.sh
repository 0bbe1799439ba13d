python scripts/ubench_thin.py df 2>&1 | grep -v Warning | tail -9
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ubp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ubp -- python $GRAFT_REPO_ROOT/scripts/ubench_thin.py df > /dev/null 2>&1
python - <<'PY'
import glob, sqlite3
db = glob.glob('/tmp/ubp/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
for n, cnt, avg in c.execute(q):
    if 'crop' in n or 'class' in n or 'fewc' in n or 'thin3' in n: print("%9.1f us x%3d  %s" % (avg, cnt, n[:70]))
PY
