#!/bin/bash
# `bench.py` at the other model sizes / object counts of BASELINE.json's configs (run on the GPU box: bash tools/bench_configs.sh);
# one line per configuration into gpurun_out/r06_bench_configs.txt.  40 timed steps, stream leg of 120 frames, no CPU baseline.
R=${GRAFT_REPO_ROOT:-.}
OUT=$R/gpurun_out/r06_bench_configs.txt
echo "# python bench.py --model M --objects B --steps 40 --warmup 3 --no-cpu-baseline --stream-frames 120 (one MI355X, bf16x3k)" > $OUT
for cfg in "sam2.1_hiera_t 4" "sam2.1_hiera_t 16" "sam2.1_hiera_s 16" "sam2.1_hiera_b+ 16" "sam2.1_hiera_l 4" "sam2.1_hiera_l 8" "sam2.1_hiera_l 16"; do
  set -- $cfg
  python $R/bench.py --model $1 --objects $2 --steps 40 --warmup 3 --no-cpu-baseline --stream-frames 120 2>/dev/null | tail -1 > /tmp/cfg.json
  python - "$1" "$2" >> $OUT <<EOF
import json, sys
d = json.loads(open("/tmp/cfg.json").read())
print("%-16s %3s objects  tracked %6.1f frames/s (%6.2f ms/frame)  stream %5.1f frames/s  cross-attention %.3f ms/launch  stages %s"
      % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["stream_fps"], d["roofline_cross_attention"]["avg_launch_ms"], d["ms_per_step_by_stage"]))
EOF
done
cat $OUT
