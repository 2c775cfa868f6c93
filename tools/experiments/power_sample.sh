#!/bin/bash
# Power / clock of the GPU while a binary runs in a loop (rocm-smi sampled every ~0.2 s): usage power_sample.sh SECONDS CMD...
# Evidence for the power-limit finding of round 4 (profiles/r04_mfma_power_calibration.txt).
secs=$1; shift
( end=$((SECONDS + secs)); while [ $SECONDS -lt $end ]; do "$@" > /dev/null 2>&1; done ) &
pid=$!
sleep 1
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk|Average Graphics" | tr -s ' \t' ' ' | tr '\n' ';'
  echo
  sleep 0.3
done
wait $pid
