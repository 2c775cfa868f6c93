#!/bin/bash
# per-kernel time of the d256 GEMM kernel for a few shapes and ablation builds (det-sam2_amd/lib/ab_<name>.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in base nodma nomfma noread; do
  for shape in "40960 2304 576" "40960 576 2304" "65536 2048 256"; do
    rm -rf /tmp/abl
    if [ $lib = base ]; then unset DS2_LIB; else export DS2_LIB=$R/det-sam2_amd/lib/ab_$lib.so; fi
    rocprofv3 --kernel-trace --stats -d /tmp/abl -o r -- python $R/tools/op_bench1.py $shape 20 > /tmp/abl.log 2>&1
    python - "$lib" "$shape" <<'PY'
import sqlite3, sys
c = sqlite3.connect('/tmp/abl/r_results.db')
rows = c.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by 3 desc").fetchall()
for n, cnt, avg, mn in rows:
    if 'gemm' in n:
        M, N, K = (int(x) for x in sys.argv[2].split())
        print(f"{sys.argv[1]:8s} {sys.argv[2]:18s} {n.split('::')[-1][:40]:40s} calls {cnt:3d} avg {avg/1e3:8.1f} us  min {mn/1e3:8.1f} us  {2.0*M*N*K/avg/1e3:7.1f} TF(alg)")
PY
  done
done
