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
# the HBM-bound kernels SURVEY 8(d) names (A3 ingest, A15 mask output, A11 bank gather, A13 first conv's input), with the
# ALGORITHMIC bytes of one launch at 16 objects / 7 bank frames / 1024^2 frames next to the counters
named = {
    "k_ingest_u8": ("A3 frame ingest (per frame: 3 MiB uint8 in, 6 MiB fp16 out)", None),
    "k_mask_output": ("A15 256^2 -> video res + threshold + bit-pack (4 MiB logits in, 2 MiB packed out)", 16 * 256 * 256 * 4 + 16 * 1024 * 128),
    "k_bank_mem": ("A11 bank gather (bf16 entries in, fp32 memory + memory_pos out)", 16 * 7 * 4096 * 64 * (2 + 4 + 4)),
    "k_mask_upsample_transform": ("A13 256^2 -> 1024^2 sigmoid*20-10 (4 MiB in, 64 MiB out; only the mask-prompt path since round 3)", 16 * 256 * 256 * 4 + 16 * 1024 * 1024 * 4),
    "k_mask_up_conv1": ("A13 256^2 logits -> upsample + sigmoid + conv3x3/s2 + LN2d + GELU (4 MiB in, 64 MiB out; the 1024^2 mask stays on chip)", 16 * 256 * 256 * 4 + 16 * 512 * 512 * 4 * 4),
}
print("# named HBM-bound kernels: measured GB/s (counters) and algorithmic GB/s (bytes a launch must move / mean duration)")
for t_ns, name, n, byt, gbs in rows:
    for key, (what, alg) in named.items():
        if key in name:
            a = f"{alg / (t_ns / n):7.0f}" if alg else "      -"
            print(f"#   {key:28s} calls {n:5d}  mean {t_ns / n / 1e3:8.1f} us  counters {gbs:6.0f} GB/s  algorithmic {a} GB/s   {what}")
