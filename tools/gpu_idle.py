#!/usr/bin/env python
"""How much of the timed region of `bench.py` is the GPU idle, and after which kernels?  (rocprofv3 --kernel-trace rocpd database)

    rocprofv3 --kernel-trace -d /tmp/idle -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stream
    python tools/gpu_idle.py /tmp/idle/r_results.db 5 20

The timed region is located by the memory cross-attention launches (4 per tracked frame): from the start of launch 4*W to the end of
the last kernel of frame W+K-1 (= the start of launch 4*(W+K), when there is one).  Busy time is the UNION of the kernel intervals over
all streams (the asynchronous encoder runs on its own stream); a gap is a stretch of the region in which no kernel of any stream runs,
attributed to the kernel that ended last before it.  This is what a hipGraph of the tracking chain could remove at most."""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    W, K = int(sys.argv[2]), int(sys.argv[3])
    rows = list(db.execute("select start, end, name from kernels order by start"))
    xa = [i for i, r in enumerate(rows) if "k_attention_x4a" in r[2]]
    assert len(xa) >= 4 * (W + K), f"only {len(xa)} cross-attention launches in the trace"
    t0 = rows[xa[4 * W]][0]
    t1 = rows[xa[4 * (W + K)]][0] if len(xa) > 4 * (W + K) else max(r[1] for r in rows)
    ks = [r for r in rows if r[1] > t0 and r[0] < t1]
    busy, cur_end, last = 0, t0, "(region start)"
    gaps = collections.defaultdict(lambda: [0, 0])
    sum_dur = 0
    for s, e, n in ks:
        s, e = max(s, t0), min(e, t1)
        sum_dur += e - s
        if s > cur_end:
            g = gaps[last]
            g[0] += 1
            g[1] += s - cur_end
            cur_end = s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
            last = n
    wall = t1 - t0
    print(f"# timed region: {K} tracked frames, {wall / 1e6:.2f} ms = {wall / 1e6 / K:.3f} ms/frame, {len(ks)} kernel launches ({len(ks) / K:.0f} per frame)")
    print(f"# union of kernel intervals {busy / 1e6:.2f} ms = {100 * busy / wall:.2f} % of the region; idle {(wall - busy) / 1e6:.2f} ms = "
          f"{(wall - busy) / 1e6 / K:.3f} ms/frame; sum of kernel durations {sum_dur / 1e6:.2f} ms ({sum_dur / wall:.3f} x the region: overlap of the encoder stream)")
    print(f"{'idle after kernel':90s} {'gaps':>6s} {'total_us':>10s} {'avg_us':>8s} {'us/frame':>9s}")
    for n, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{n[:90]:90s} {c:6d} {t / 1e3:10.1f} {t / 1e3 / c:8.2f} {t / 1e3 / K:9.2f}")


if __name__ == "__main__":
    main()
