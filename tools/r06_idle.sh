#!/bin/bash
# GPU idle share of the timed region (tools/gpu_idle.py) for the headline shape and for BASELINE config 2 (GPU box).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_gpu_idle.txt
: > $OUT
for cfg in "sam2.1_hiera_l 16" "sam2.1_hiera_t 4"; do
  set -- $cfg
  rm -rf /tmp/idle
  timeout 600 rocprofv3 --kernel-trace -d /tmp/idle -o r -- python $R/bench.py --model $1 --objects $2 --steps 20 --warmup 5 --no-cpu-baseline --no-stream --no-hole-filling-leg > /tmp/idle.log 2>&1 || tail -5 /tmp/idle.log
  echo "## $1, $2 objects (asynchronous encoder on, rocprofv3 --kernel-trace): $(grep '^{' /tmp/idle.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("bench line under the tracer %.1f frames/s" % d["value"])')" >> $OUT
  python $R/tools/gpu_idle.py /tmp/idle/r_results.db 5 20 >> $OUT 2>&1
done
cat $OUT | cut -c1-200
