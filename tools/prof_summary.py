#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table we commit under profiles/.

    python tools/prof_summary.py gpurun_out/prof_r01/bench_l_results.db > profiles/r01_xxx_kernel_stats.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute(
    "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), "
    "max(accum_vgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary ({sys.argv[1]}); total kernel time {tot / 1e6:.2f} ms")
print(f"{'kernel':95s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'scratch':>7s}")
for r in rows:
    print(f"{r[0][:95]:95s} {r[1]:7d} {r[2] / 1e6:10.2f} {r[3] / 1e3:10.1f} {r[4] / 1e3:9.1f} {r[5] / 1e3:9.1f} {100 * r[2] / tot:6.2f} "
          f"{r[6] or 0:5d} {r[7] or 0:5d} {r[8] or 0:7d} {r[9] or 0:7d}")
