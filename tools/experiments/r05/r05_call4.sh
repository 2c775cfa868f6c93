echo "=== default build (e3 with residual loads one step ahead): parity + timing"
timeout 400 python tools/x4g_check.py big 5 2>&1 | grep -v amdgpu.ids | cut -c1-420
echo "=== nostore"
DS2_LIB=det-sam2_amd/lib/ab_nostore.so timeout 300 python tools/x4g_check.py big 5 --nocheck 2>&1 | grep -v amdgpu.ids | sed -e 's/bit-identical //g' | cut -c1-400
for gm in 0 2 4 16; do
  echo "=== DS2_GEMM_GROUPM=$gm"
  DS2_GEMM_GROUPM=$gm timeout 300 python tools/x4g_check.py big 5 --nocheck 2>&1 | grep -v amdgpu.ids | sed -e 's/bit-identical //g' | cut -c1-400 | head -4
done
cd /tmp && export TMPDIR=/tmp
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/pmcx
  timeout 600 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pmcx -o r -- python $GRAFT_REPO_ROOT/tools/x4g_check.py big 3 --nocheck > /tmp/pmcx.log 2>&1 || tail -3 /tmp/pmcx.log
  python $GRAFT_REPO_ROOT/tools/pmc_all_parse.py /tmp/pmcx/r_results.db $pass | grep -v "split_rows\|elementwise\|native"
done
