"""Join the FETCH_SIZE and WRITE_SIZE PMC passes per kernel name -> achieved HBM GB/s table (see pmc_hbm_table.sh)."""
import sqlite3
import sys
from collections import defaultdict


def load(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, dispatch_id, sum(value), max(duration) from counters_collection where counter_name = ? "
                     "group by dispatch_id", (counter,)).fetchall()
    agg = defaultdict(lambda: [0.0, 0.0, 0])
    for name, _, v, dur in rows:
        a = agg[name]
        a[0] += v
        a[1] += dur
        a[2] += 1
    return agg


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for name in set(f) | set(w):
    fk, ft, fn = f.get(name, [0, 0, 0])
    wk, wt, wn = w.get(name, [0, 0, 0])
    t_ns = (ft + wt) / 2 if ft and wt else (ft or wt)
    n = max(fn, wn)
    byt = (2.0 * fk + wk) * 1024.0
    if t_ns > 0:
        rows.append((t_ns, name, n, byt, byt / t_ns))
rows.sort(reverse=True)
tot_t = sum(r[0] for r in rows)
tot_b = sum(r[3] for r in rows)
print("# achieved HBM traffic per kernel: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, time = dispatch durations of the PMC runs")
print(f"# all kernels: {tot_b / 1e9:.1f} GB in {tot_t / 1e6:.1f} ms of kernel time = {tot_b / tot_t:.0f} GB/s average (peak 8000, achievable ~6300)")
print(f"{'kernel':80s} {'calls':>6s} {'time_ms':>9s} {'GB':>8s} {'GB/s':>7s}")
for t_ns, name, n, byt, gbs in rows[:40]:
    print(f"{name[:80]:80s} {n:6d} {t_ns / 1e6:9.2f} {byt / 1e9:8.2f} {gbs:7.0f}")
