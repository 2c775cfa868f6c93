for v in k_std; do
  echo "=== $v"; DS2_LIB=det-sam2_amd/lib/ab_$v.so timeout 300 python tools/x4g_check.py big 5 --only 0,2,3,6 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
echo "=== default"; timeout 300 python tools/x4g_check.py big 5 --only 0,2,3,6 2>&1 | grep -v amdgpu.ids | cut -c1-330
