"""Per-kernel sums of every PMC counter in a rocprofv3 results .db (dev aid)."""
import collections, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for k, c, v in cur.execute("select kernel_name,counter_name,value from counters_collection"):
    agg[k][c] += v
names = sorted({c for a in agg.values() for c in a})
for k in sorted(agg, key=lambda k: -max(agg[k].values()))[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(k[:70]); print("   " + "  ".join("%s=%.4g" % (c, agg[k][c]) for c in names))
