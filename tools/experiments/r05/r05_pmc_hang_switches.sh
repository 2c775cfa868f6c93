cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; export DS2_ASYNC_ENCODE=0
for v in "DS2_MA_VOFUSE=0" "DS2_MA_QKVFUSE=0 DS2_MA_QFUSE=0" "DS2_BANK_DIRECT=0" "X=1"; do
  rm -rf /tmp/pmc_B
  s=$(date +%s)
  env $v timeout 60 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_B -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --stream-frames 60 > /tmp/pmc_B.log 2>&1
  rc=$?
  echo "$v rc=$rc $(( $(date +%s) - s )) s"
done
