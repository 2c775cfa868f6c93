#!/bin/bash
# HBM table only (two PMC passes of the bench incl. its stream leg; rocprofv3 segfaults sporadically with --pmc here: retried)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DS2_ASYNC_ENCODE=0
for c in FETCH_SIZE WRITE_SIZE; do
  for try in 1 2 3; do
    rm -rf /tmp/hbm_$c
    rocprofv3 --pmc $c --kernel-trace -d /tmp/hbm_$c -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --stream-frames 40 > /tmp/hbm_$c.log 2>&1 && break
    echo "pass $c try $try failed"
  done
done
python $R/tools/pmc_hbm_table.py /tmp/hbm_FETCH_SIZE/r_results.db /tmp/hbm_WRITE_SIZE/r_results.db > $R/gpurun_out/r03_hbm_by_kernel.txt
tail -8 $R/gpurun_out/r03_hbm_by_kernel.txt | cut -c1-200
