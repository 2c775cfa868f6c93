#!/bin/bash
# Achieved HBM GB/s per kernel of the bench (north star: "rocprof must show achieved HBM GB/s on the encoder"):
# two PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of `bench.py --steps 4`, joined per kernel name.
# bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB (gfx950 read-side correction, MI355X_MICROARCH.md); time = the dispatch
# durations of the same (profiled) runs.  Run on the GPU box: bash tools/pmc_hbm_table.sh > gpurun_out/hbm_table.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/hbm_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/hbm_$c -o r -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /tmp/hbm_$c.log 2>&1 || tail -5 /tmp/hbm_$c.log
done
python $R/tools/pmc_hbm_table.py /tmp/hbm_FETCH_SIZE/r_results.db /tmp/hbm_WRITE_SIZE/r_results.db
