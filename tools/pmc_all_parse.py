#!/usr/bin/env python
"""Per-(kernel, grid) means of the PMC counters of one rocprofv3 pass (rocpd sqlite)."""
import re
import sqlite3
import sys
from collections import defaultdict

db, counters = sys.argv[1], sys.argv[2:]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
gcol = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
q = f"select dispatch_id, kernel_name, {gcol or '0'}, counter_name, sum(value), max(duration) from counters_collection group by dispatch_id, counter_name"
per = defaultdict(dict)
meta = {}
for did, kn, grid, cn, v, du in c.execute(q):
    per[did][cn] = v
    meta[did] = (kn, grid, du)
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
dur = defaultdict(float)
for did, vals in per.items():
    kn, grid, du = meta[did]
    mo = re.search(r"(k_\w+(?:<[^>]*>)?|__amd_\w+|at::native::\w+)", kn)
    key = ((mo.group(1) if mo else kn)[:60], grid)
    cnt[key] += 1
    dur[key] += du
    for cn, v in vals.items():
        agg[key][cn] += v
print("# one rocprofv3 --pmc pass (" + " ".join(counters) + "), bench.py --steps 4; means per launch; SQ_*_CYCLES in the units of MI355X_MICROARCH.md")
print(f"{'kernel':60s} {'grid':>9s} {'calls':>6s} {'avg_us':>9s} " + " ".join(f"{x[-22:]:>22s}" for x in counters))
for key in sorted(dur, key=lambda k: -dur[k])[:40]:
    n = cnt[key]
    print(f"{key[0]:60s} {key[1]:9d} {n:6d} {dur[key] / n / 1e3:9.1f} " + " ".join(f"{agg[key].get(x, 0) / n:22.0f}" for x in counters))
