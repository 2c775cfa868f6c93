#!/bin/bash
# kernel time of one GEMM shape under several forced tiles (rocprofv3 kernel trace):  bash tools/gemm_tile_time.sh "5 9" "65536 2304 576" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TILES=$1; shift
for shape in "$@"; do
  for t in $TILES; do
    rm -rf /tmp/gtt
    DS2_GEMM_TILE=$t rocprofv3 --kernel-trace --stats -d /tmp/gtt -o r -- python $R/tools/op_bench1.py $shape 20 > /tmp/gtt.log 2>&1
    python - "$t" "$shape" <<'PY'
import sqlite3, sys
c = sqlite3.connect('/tmp/gtt/r_results.db')
rows = c.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by 3 desc").fetchall()
for n, cnt, avg, mn in rows:
    if 'gemm' in n:
        M, N, K = (int(x) for x in sys.argv[2].split())
        print(f"tile {sys.argv[1]:3s} {sys.argv[2]:18s} {n.split('::')[-1][:40]:40s} calls {cnt:3d} avg {avg/1e3:8.1f} us  min {mn/1e3:8.1f} us  {2.0*M*N*K/avg/1e3:7.1f} TF(alg)")
PY
  done
done
