#!/bin/bash
# Round-3 profile set (run on the GPU box: bash tools/r03_profile.sh): kernel-trace stats of the bench with the async
# encoder off (every duration = the kernel alone), the per-kernel HBM GB/s table (incl. the stream leg: ingest, mask output),
# and the PMC traffic of the dominant kernels.  Everything lands in gpurun_out/r03_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DS2_ASYNC_ENCODE=0
rm -rf /tmp/prof_r03
rocprofv3 --kernel-trace --stats -d /tmp/prof_r03 -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stream > /tmp/prof_r03.log 2>&1 || tail -5 /tmp/prof_r03.log
python $R/tools/prof_summary.py /tmp/prof_r03/r_results.db > $R/gpurun_out/r03_bench_l_bf16x3k_noasync_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/hbm_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/hbm_$c -o r -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --stream-frames 60 > /tmp/hbm_$c.log 2>&1 || tail -5 /tmp/hbm_$c.log
done
python $R/tools/pmc_hbm_table.py /tmp/hbm_FETCH_SIZE/r_results.db /tmp/hbm_WRITE_SIZE/r_results.db > $R/gpurun_out/r03_hbm_by_kernel.txt
# traffic of the dominant kernels at the BENCH workload (Nk = 28736: no stream leg, whose bank holds up to 3 conditioning frames)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o r -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stream > /tmp/pmc_$c.log 2>&1 || tail -5 /tmp/pmc_$c.log
done
python $R/tools/pmc_traffic_parse.py /tmp/pmc_FETCH_SIZE/r_results.db /tmp/pmc_WRITE_SIZE/r_results.db > $R/gpurun_out/r03_pmc_traffic.json
head -12 $R/gpurun_out/r03_bench_l_bf16x3k_noasync_kernel_stats.txt | cut -c1-180
head -8 $R/gpurun_out/r03_hbm_by_kernel.txt | cut -c1-160
cat $R/gpurun_out/r03_pmc_traffic.json | head -30
