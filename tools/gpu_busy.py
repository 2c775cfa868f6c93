#!/usr/bin/env python
"""GPU busy fraction of the steady state of a kernel trace (rocpd sqlite): over the last FRAC of the dispatches, the union of the
kernel intervals / the wall span - how much of a tracked frame the GPU waits for the host (launch-bound shapes).

    python tools/gpu_busy.py /tmp/ks_NAME/r_results.db [FRAC=0.5]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(db.execute("select start, end from kernels order by start"))
rows = rows[int(len(rows) * (1 - frac)):]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
gaps.sort()
print(f"{len(rows)} dispatches, span {span / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms = {100.0 * busy / span:.1f} %; "
      f"{len(gaps)} gaps, median {gaps[len(gaps) // 2] / 1e3:.1f} us, mean {sum(gaps) / len(gaps) / 1e3:.1f} us, "
      f"sum {sum(gaps) / 1e6:.2f} ms, the 10 largest {[round(g / 1e3) for g in gaps[-10:]]} us")
