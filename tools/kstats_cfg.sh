#!/bin/bash
# Kernel-trace stats of a short bench run of another configuration, async encoder off (run on the GPU box):
#   bash tools/kstats_cfg.sh NAME MODEL OBJECTS [steps]   -> gpurun_out/NAME_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DS2_ASYNC_ENCODE=0
rm -rf /tmp/ks_$1
rocprofv3 --kernel-trace --stats -d /tmp/ks_$1 -o r -- python $R/bench.py --model $2 --objects $3 --steps ${4:-16} --warmup 2 --no-cpu-baseline --no-stream > /tmp/ks_$1.log 2>&1 || tail -5 /tmp/ks_$1.log
python $R/tools/prof_summary.py /tmp/ks_$1/r_results.db > $R/gpurun_out/$1_kernel_stats.txt
