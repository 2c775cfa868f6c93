#!/bin/bash
# Variant of the generated MLP loop (tools/gen/gen_mlp256_x4m.py) as an A/B library:  bash tools/x4m_variant.sh NAME [X4M_D=6] [X4M_FLAGS="nobar noread"]
# -> det-sam2_amd/lib/ab_NAME.so (time it with tools/mlp_time.py NAME; ablation flags give WRONG results by construction)
set -e
name=$1; shift
mkdir -p /tmp/x4m_var
rm -f /tmp/x4m_var/$name.inc det-sam2_amd/lib/ab_$name.so
env "$@" python tools/gen/gen_mlp256_x4m.py /tmp/x4m_var/$name.inc
python tools/ab.py build $name "-DX4M_INC_FILE=\"/tmp/x4m_var/$name.inc\"" | tail -1
