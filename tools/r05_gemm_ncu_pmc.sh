#!/bin/bash
# VERDICT r4 next #1, first step: is the 1.5x-per-CU speed-up of k_gemm_split_pp256 on fewer CUs clock (power) or memory-system
# contention?  The same 65536 x 2304 x 576 product on DS2_GEMM_NCU = 256 / 128 / 64 persistent workgroups, one PMC pass each:
# clock = SQ_BUSY_CYCLES / (SQ instances of the busy XCDs' CUs) / duration; matrix pipe share = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES.
# usage (GPU box): bash tools/r05_gemm_ncu_pmc.sh > gpurun_out/r05_gemm_ncu_pmc.txt
cd /tmp && export TMPDIR=/tmp
for NCU in 256 128 64; do
  rm -rf /tmp/pmcn
  DS2_GEMM_X4G=0 DS2_GEMM_NCU=$NCU timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmcn -o r -- \
    python $GRAFT_REPO_ROOT/tools/experiments/gemm_ncu.py > /tmp/pmcn.log 2>&1 || tail -5 /tmp/pmcn.log
  python - "$NCU" <<'PY'
import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect('/tmp/pmcn/r_results.db')
rows = c.execute("select dispatch_id, counter_name, sum(value), max(duration) from counters_collection where kernel_name like '%k_gemm_split_pp256%' group by dispatch_id, counter_name").fetchall()
d = defaultdict(dict); dur = {}
for did, cn, v, du in rows: d[did][cn] = v; dur[did] = du
ks = sorted(d)[10:]          # skip warm-up launches
n = len(ks)
agg = defaultdict(float)
for k in ks:
    for cn, v in d[k].items(): agg[cn] += v / n
du = sum(dur[k] for k in ks) / n / 1e3
ncu = int(sys.argv[1])
busy, mfma = agg.get('SQ_BUSY_CYCLES', 0), agg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
print(f"NCU={ncu:3d}: {n} launches, avg {du:8.1f} us under the counters | SQ_BUSY_CYCLES {busy:.3e}  SQ_VALU_MFMA_BUSY {mfma:.3e}  SQ_WAIT_ANY {agg.get('SQ_WAIT_ANY',0):.3e}  SQ_WAVE_CYCLES {agg.get('SQ_WAVE_CYCLES',0):.3e}")
print(f"         MFMA_BUSY/BUSY = {mfma / max(busy,1):.2f} (x / 4 SIMDs... see profiles/r04_pmc_by_kernel_sq.txt for the normalisation: / 32 SQ instances);  "
      f"clock = BUSY / 32 / t = {busy / 32 / du / 1e3:.3f} GHz;  wait share = WAIT_ANY / WAVE_CYCLES = {agg.get('SQ_WAIT_ANY',0) / max(agg.get('SQ_WAVE_CYCLES',1),1):.2f}")
PY
done
