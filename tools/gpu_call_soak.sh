#!/usr/bin/env bash
# differential soaks through the PRODUCT library on the GPU box (tests/soak.py, SOAK_LIB=hip): many more seeds than the suite runs
set -u
OUT=gpurun_out/${TAG:-r03z2}; mkdir -p "$OUT"
export TMPDIR=/tmp
for k in "preempt 1500" "rounds 2000" "streams 500" "market 1500" "optimiser 800" "away 800" "offgrid 800" "features 400" "ops 3000" "fit 400"; do
  SOAK_LIB=hip timeout 900 python tests/soak.py $k 2>&1 | tail -1 | tee -a "$OUT/soak_hip.txt"
done
