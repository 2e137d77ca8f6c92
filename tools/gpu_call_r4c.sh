#!/usr/bin/env bash
# round 4, third call: wide runs (more than 64 queues) on the MI355X — parity tests, the 256 / 1024-queue rounds timed, and the headline A/B against round 3's library
set -u
OUT=gpurun_out/r04c; mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu -k "wide or comm_in_library" -p no:cacheprovider --durations=8 > $OUT/pytest_wide.log 2>&1; echo "wide pytest rc=$?" | tee -a $OUT/summary.txt; tail -n 14 $OUT/pytest_wide.log | tee -a $OUT/summary.txt
for W in q256 q1024 q64; do
  echo "== new $W" >> $OUT/summary.txt
  timeout 300 python tools/prof_config4.py $W 2>&1 | tail -n 2 >> $OUT/summary.txt
done
echo "== new q256 ASCHED_WIDE=0 (generic path)" >> $OUT/summary.txt
ASCHED_WIDE=0 timeout 300 python tools/prof_config4.py q256 2>&1 | tail -n 1 >> $OUT/summary.txt
bash tools/ab_call.sh r04c "base new" "headline gangs preempt" ""
cat $OUT/ab.txt >> $OUT/summary.txt
