#!/usr/bin/env bash
# round 3, call c: the fair-share threshold table (round_ft.h) — parity tests, A/B against the round-2 build, full-size configs[4], segment profile
set -u
OUT=gpurun_out/r3c; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "at_scale or 64k or preempt or order_key or pqs_goldens or crowded or fast_path or random_rounds" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | tee -a $OUT/summary.txt
bash tools/ab_call.sh r3c "base new" "headline gangs preempt" ""
for ft in 1 0; do
  echo "== full-size configs[4], ASCHED_FT=$ft" | tee -a $OUT/full4.txt
  ASCHED_FT=$ft timeout 600 python tools/prof_config4.py full 2>&1 | tail -n 1 | tee -a $OUT/full4.txt
done
for ft in 1 0; do
  echo "== segments (profiling build), 20k x 200k preemption-heavy, ASCHED_FT=$ft" | tee -a $OUT/segments.txt
  ASCHED_FT=$ft ASCHED_LIB_PATH=$PWD/armada_amd/csrc/libarmada_sched_prof.so ASCHED_PRINT_SEG=1 timeout 300 python tools/prof_config4.py 2>&1 | tail -n 2 | tee -a $OUT/segments.txt
done
