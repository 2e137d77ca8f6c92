#!/usr/bin/env bash
set -u
OUT=gpurun_out/r3g; mkdir -p $OUT
export ASCHED_SAFETY_DEADLINE_S=150
bash tools/ab_call.sh r3g "base new" "headline gangs preempt" "at_scale or 64k or preempt or goldens or crowded or fast_path"
for ft in 1 0; do
  echo "== full-size configs[4] FT=$ft" | tee -a $OUT/full4.txt
  ASCHED_FT=$ft timeout 600 python tools/prof_config4.py full 2>&1 | tail -n 1 | tee -a $OUT/full4.txt
done
