#!/bin/bash
# per-layer-group sweep of two-MFMA-equivalent products in the IMAGE ENCODER (CPU, oracle emulation).
# usage: bash tools/r05_prec_sweep.sh <scheme> <case> [groups...]   -> appends to gpurun_out/r05_prec_sweep.txt
SCHEME=$1; CASE=$2; shift 2
GRPS=${@:-s12 s3qkv s3proj s3fc1 s3fc2 s4 neck}
for g in $GRPS; do
  r=$(DS2_EMU_ONLY=enc:$g DS2_EMU_THREADS=${DS2_EMU_THREADS:-2} python tools/prec_emulate.py $SCHEME $CASE 2>&1 | tail -1)
  echo "enc:$g $r" >> gpurun_out/r05_prec_sweep.txt
done
