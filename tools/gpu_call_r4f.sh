#!/usr/bin/env bash
set -u
OUT=gpurun_out/r04f; mkdir -p $OUT
timeout 600 python tools/probe_refbench.py > $OUT/refbench.txt 2>&1; tail -n 9 $OUT/refbench.txt | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -q -m gpu -k "away or goldens or random_rounds or feature_mix or stream_runs or wide" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -n 3 $OUT/pytest.log | tee -a $OUT/summary.txt
for H in 127 191 255; do
  echo "== config4 full, ASCHED_HELPERS=$H" | tee -a $OUT/summary.txt
  ASCHED_HELPERS=$H timeout 400 python tools/prof_config4.py full 2>&1 | tail -n 1 | cut -c1-330 | tee -a $OUT/summary.txt
done
