timeout 900 python -m pytest tests/test_hip_stages.py -q -x -k "key_projection or bench_size" 2>&1 | tail -4
python tools/ab.py env base nok64t:DS2_GEMM_K64T=0 --rounds 2 2>&1 | grep -v amdgpu.ids | cut -c1-260
