#!/usr/bin/env bash
# round 4: segment clocks of the wide runs (256 / 1024 queues)
set -u
OUT=gpurun_out/r04d; mkdir -p $OUT
for W in q256 q1024; do
  echo "== $W" >> $OUT/summary.txt
  ASCHED_PRINT_SEG=1 timeout 300 python tools/prof_config4.py $W 2>&1 | tail -n 3 >> $OUT/summary.txt
done
cat $OUT/summary.txt
