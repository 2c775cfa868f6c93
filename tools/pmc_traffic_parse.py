"""FETCH_SIZE / WRITE_SIZE (KB, rocprofv3 derived counters) of the cross-attention launches with the full bank ->
JSON for bench.py.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
coalesced read stream (128-byte requests tallied at 64 B) => the read side is doubled; WRITE_SIZE is taken as is."""
import json
import sqlite3
import sys


def per_dispatch(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, sum(value), max(duration) from counters_collection where kernel_name like "
                     "'%k_attention_w8<64%' and counter_name = ? group by dispatch_id", (counter,)).fetchall()
    return rows


f = per_dispatch(sys.argv[1], "FETCH_SIZE")
w = per_dispatch(sys.argv[2], "WRITE_SIZE")


def steady(rows):   # launches with the full 7-frame bank = the longest ones
    dmax = max(r[2] for r in rows)
    sel = [r[1] for r in rows if r[2] >= 0.9 * dmax]
    return sum(sel) / len(sel), len(sel)


fk, nf = steady(f)
wk, nw = steady(w)
out = {"kernel": "k_attention_w8<64,2,*> (memory cross-attention, Nk=28736, 16 objects; default arithmetic mode of the run)",
       "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk, "launches_averaged": [nf, nw],
       "read_correction": 2.0,
       "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on "
                 "`python bench.py --steps 4 --warmup 1 --no-cpu-baseline`; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024"}
print(json.dumps(out))
