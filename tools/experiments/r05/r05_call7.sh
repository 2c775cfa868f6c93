for v in g16 g32 g40; do
  echo "=== $v"; DS2_LIB=det-sam2_amd/lib/ab_$v.so timeout 300 python tools/x4g_check.py big 5 --nocheck 2>&1 | grep -v amdgpu.ids | sed -e 's/bit-identical //g' | cut -c1-330 | head -4
done
echo "=== default"; timeout 300 python tools/x4g_check.py big 5 2>&1 | grep -v amdgpu.ids | cut -c1-330
timeout 600 python -m pytest tests/test_hip_gemm_x4g.py -x -q 2>&1 | tail -5
python tools/ab.py env base nox4g:DS2_GEMM_X4G=0 --rounds 2 2>&1 | grep -v amdgpu.ids
