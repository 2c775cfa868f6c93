"""Print per-kernel PMC counter values from a rocprofv3 rocpd sqlite database (one row per dispatch)."""
import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else '%'
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
pm = [t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower()]
if '--schema' in sys.argv:
    for t in pm:
        print(t, [r[1] for r in c.execute(f"pragma table_info({t})")])
    sys.exit()
view = 'counters_collection' if 'counters_collection' in tabs else pm[0]
cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
rows = c.execute(f"select dispatch_id, kernel_name, grid_size, counter_name, sum(value) from {view} where kernel_name like ? group by dispatch_id, counter_name order by dispatch_id", (pat,)).fetchall()
d = defaultdict(dict)
for did, kn, gs, cn, v in rows:
    d[(did, kn[:40], gs)][cn] = v
for k, v in d.items():
    print(k, {a: round(b) for a, b in v.items()})
