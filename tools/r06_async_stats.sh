#!/bin/bash
# Kernel table of the HEADLINE command's shape with the asynchronous encoder ON (the noasync table is the one the per-kernel shares
# come from; this one is for comparing `roofline.avg_launch_ms` of the live bench line with the tracer's average).  GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_async
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_async -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stream --no-hole-filling-leg > /tmp/async.log 2>&1 || tail -5 /tmp/async.log
OUT=$R/gpurun_out/r06_bench_l_bf16x3k_async_kernel_stats.txt
echo "# python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stream --no-hole-filling-leg under rocprofv3 --kernel-trace --stats (asynchronous encoder ON)" > $OUT
echo "# bench line of this run: $(grep '^{' /tmp/async.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.2f frames/s, %.3f ms/frame, roofline.avg_launch_ms %.4f (HIP events, live), frac %.3f" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"]))')" >> $OUT
python $R/tools/prof_summary.py /tmp/prof_async/r_results.db >> $OUT
head -12 $OUT | cut -c1-200
