#!/bin/bash
# usage (on the GPU box): bash tools/gemm_ablate.sh ; prints k_gemm_split* times per shape for each ablation mask
cd /tmp && export TMPDIR=/tmp
for dbg in ${DBGS:-0 1 2 3 4 8 12}; do
  export DS2_GEMM_DBG=$dbg
  rm -rf /tmp/p_$dbg
  rocprofv3 --kernel-trace -d /tmp/p_$dbg -o r -- python $GRAFT_REPO_ROOT/tools/gemm_ablate.py > /tmp/abl_$dbg.log 2>&1 || tail -5 /tmp/abl_$dbg.log
  echo "DBG=$dbg"
  python $GRAFT_REPO_ROOT/tools/gemm_ablate_parse.py /tmp/p_$dbg/r_results.db
done
