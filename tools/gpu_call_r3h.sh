#!/usr/bin/env bash
set -u
OUT=gpurun_out/r3h; mkdir -p $OUT
export ASCHED_SAFETY_DEADLINE_S=200
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -18 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
bash tools/ab_call.sh r3h "base new" "headline gangs preempt" ""
