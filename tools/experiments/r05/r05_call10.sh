for v in d0 d_nobar d_a d_w d_same d_nt full_same full_nt; do
  echo "=== $v"; DS2_LIB=det-sam2_amd/lib/ab_$v.so timeout 300 python tools/x4g_check.py big 5 --nocheck --only 0,1,3 2>&1 | grep -v amdgpu.ids | sed -e 's/bit-identical //g' | cut -c1-330
done
