import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else [x for x in cols if "grid" in x][0]
rows = c.execute(f"select name, {gx}, (end-start) from kernels where name like '%k_gemm%' order by start").fetchall()
d = defaultdict(list)
for n, g, t in rows:
    d[(n[:34], g)].append(t / 1e3)
for g, v in d.items():
    print("  ", g, "us", [round(x, 1) for x in v])
if not rows:
    print("  no rows; columns:", cols)
