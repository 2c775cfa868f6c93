python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-stream --gemm-table gpurun_out/r05_h_gemm_table_x4g.txt > gpurun_out/r05_h_bench_x4g.json 2>gpurun_out/r05_h_err1.txt
DS2_GEMM_X4G=0 python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-stream --gemm-table gpurun_out/r05_h_gemm_table_nox4g.txt > gpurun_out/r05_h_bench_nox4g.json 2>gpurun_out/r05_h_err2.txt
grep -A12 "^# k_gemm_x4g\|^# k_gemm_split_pp256\|^# k_gemm_split_r3" gpurun_out/r05_h_gemm_table_x4g.txt | cut -c1-110 | head -80
echo ======== nox4g
grep -A12 "^# k_gemm_split_pp256\|^# k_gemm_split_r3" gpurun_out/r05_h_gemm_table_nox4g.txt | cut -c1-110 | head -50
