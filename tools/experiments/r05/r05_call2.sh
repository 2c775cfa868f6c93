timeout 120 python tools/x4g_check.py small > gpurun_out/r05_b_x4g_small.txt 2>&1; echo "small rc=$?" >> gpurun_out/r05_b_x4g_small.txt
head -c 6000 gpurun_out/r05_b_x4g_small.txt
if grep -q "X4G CHECK PASS" gpurun_out/r05_b_x4g_small.txt; then
  timeout 400 python tools/x4g_check.py big 5 > gpurun_out/r05_b_x4g_big.txt 2>&1; echo "big rc=$?" >> gpurun_out/r05_b_x4g_big.txt
  tail -12 gpurun_out/r05_b_x4g_big.txt
fi
