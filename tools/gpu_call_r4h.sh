#!/usr/bin/env bash
set -u
OUT=gpurun_out/r04h; mkdir -p $OUT
for W in full ""; do
  echo "== trace config4 $W" | tee -a $OUT/summary.txt
  ASCHED_LIB_PATH=$PWD/armada_amd/csrc/libarmada_sched_trace.so timeout 400 python tools/prof_config4.py $W 2>&1 | tail -n 5 | cut -c1-400 | tee -a $OUT/summary.txt
done
