#!/bin/bash
# HBM traffic of the dominant kernel (memory cross-attention) for bench.py's roofline.traffic.
# Two separate PMC passes (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2), --kernel-trace only, same bench
# command.  Run on the GPU box:  bash tools/pmc_traffic.sh ; writes gpurun_out/pmc_cross_attention.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  DS2_ASYNC_ENCODE=0 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o r -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stream > /tmp/pmc_$c.log 2>&1 || tail -5 /tmp/pmc_$c.log
  ls /tmp/pmc_$c | head -3; tail -2 /tmp/pmc_$c.log | cut -c1-300
done
python $R/tools/pmc_traffic_parse.py /tmp/pmc_FETCH_SIZE/r_results.db /tmp/pmc_WRITE_SIZE/r_results.db > $R/gpurun_out/pmc_cross_attention.json
cat $R/gpurun_out/pmc_cross_attention.json
