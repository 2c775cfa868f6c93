#!/bin/bash
# usage: bash tools/pmc_kernel.sh "<kernel name LIKE pattern>" COUNTER [COUNTER...]   (one PMC pass of bench.py --steps 2)
cd /tmp && export TMPDIR=/tmp
PAT="$1"; shift
rm -rf /tmp/pmck
rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmck -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmck.log 2>&1 || tail -5 /tmp/pmck.log
python - "$PAT" <<'PY'
import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect('/tmp/pmck/r_results.db')
rows = c.execute("select dispatch_id, counter_name, sum(value), max(duration) from counters_collection where kernel_name like ? group by dispatch_id, counter_name", (sys.argv[1],)).fetchall()
d = defaultdict(dict); dur = {}
for did, cn, v, du in rows: d[did][cn] = v; dur[did] = du
if not d: print("no dispatch matches"); sys.exit()
dmax = max(dur.values())
sel = [k for k in d if dur[k] >= 0.9 * dmax]
agg = defaultdict(float)
for k in sel:
    for cn, v in d[k].items(): agg[cn] += v / len(sel)
print(f"{len(sel)} launches, avg duration {sum(dur[k] for k in sel)/len(sel)/1e3:.1f} us")
for cn, v in sorted(agg.items()): print(f"  {cn:32s} {v:16.0f}")
PY
