#!/bin/bash
# usage: bash tools/pmc_all.sh OUT.txt COUNTER [COUNTER...]     (ONE rocprofv3 PMC pass of `bench.py --steps 4`, --kernel-trace
# only - never combined with other trace domains).  Writes, per (kernel, grid size), the mean of every counter per launch.
cd /tmp && export TMPDIR=/tmp
OUT="$1"; shift
rm -rf /tmp/pmca
timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmca -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stream > /tmp/pmca.log 2>&1 || tail -5 /tmp/pmca.log
python $GRAFT_REPO_ROOT/tools/pmc_all_parse.py /tmp/pmca/r_results.db "$@" > "$GRAFT_REPO_ROOT/$OUT"
