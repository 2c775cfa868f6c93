timeout 120 python tools/x4g_check.py small 2>&1 | grep -v amdgpu.ids | cut -c1-900 | head -60
timeout 400 python tools/x4g_check.py big 5 2>&1 | grep -v amdgpu.ids | cut -c1-420
