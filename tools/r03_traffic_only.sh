cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DS2_ASYNC_ENCODE=0
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o r -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stream > /tmp/pmc_$c.log 2>&1 || tail -5 /tmp/pmc_$c.log
done
python $R/tools/pmc_traffic_parse.py /tmp/pmc_FETCH_SIZE/r_results.db /tmp/pmc_WRITE_SIZE/r_results.db > $R/gpurun_out/r03_pmc_traffic.json
cat $R/gpurun_out/r03_pmc_traffic.json
