for v in base nodrain nodma noread nomfma nobar; do
  echo "=== $v"; DS2_LIB=det-sam2_amd/lib/ab_$v.so timeout 300 python tools/x4g_check.py big 5 --nocheck 2>&1 | grep -v amdgpu.ids | sed -e 's/bit-identical //g' | cut -c1-400
done
