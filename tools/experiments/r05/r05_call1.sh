set -x
timeout 120 python tools/x4g_check.py small > gpurun_out/r05_a_x4g_small.txt 2>&1; echo "small rc=$?" >> gpurun_out/r05_a_x4g_small.txt
tail -15 gpurun_out/r05_a_x4g_small.txt
if grep -q "X4G CHECK" gpurun_out/r05_a_x4g_small.txt; then
  timeout 400 python tools/x4g_check.py big 5 > gpurun_out/r05_a_x4g_big.txt 2>&1; echo "big rc=$?" >> gpurun_out/r05_a_x4g_big.txt
  tail -12 gpurun_out/r05_a_x4g_big.txt
fi
bash tools/r05_gemm_ncu_pmc.sh > gpurun_out/r05_gemm_ncu_pmc.txt 2>&1
cat gpurun_out/r05_gemm_ncu_pmc.txt
