timeout 120 python tools/x4g_check.py small 2>&1 | grep -v amdgpu.ids | cut -c1-300 | head -30
timeout 400 python tools/x4g_check.py big 5 2>&1 | grep -v amdgpu.ids | cut -c1-330
echo "=== 3 stages for cfg 23 (the previous ring)"
DS2_LIB=det-sam2_amd/lib/ab_s3.so timeout 300 python tools/x4g_check.py big 5 --nocheck 2>&1 | grep -v amdgpu.ids | sed -e 's/bit-identical //g' | cut -c1-330
