#!/usr/bin/env bash
# round 3, call d: XCD-local helper workgroups (ASCHED_HELPERS_LOCAL) x fair-share threshold table (ASCHED_FT): sanity, parity tests, A/B matrix
set -u
OUT=gpurun_out/r3d; mkdir -p $OUT
export ASCHED_SAFETY_DEADLINE_S=150
ASCHED_PRINT_HELPERS=1 timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt; grep -m4 "asched helpers" $OUT/smoke.log | tee -a $OUT/summary.txt
ASCHED_FT=1 timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "at_scale or 64k or preempt or order_key or goldens or crowded or fast_path or random_rounds or stream or timeout or split or gang" > $OUT/pytest.log 2>&1; echo "pytest(local,ft) rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | tee -a $OUT/summary.txt
for loc in 1 0; do
  echo "== headline ASCHED_HELPERS_LOCAL=$loc" | tee -a $OUT/matrix.txt
  ASCHED_HELPERS_LOCAL=$loc timeout 600 python bench.py --steps 5 --warmup 1 --cpu-budget 0 --no-other 2>/dev/null | head -c 300 | tee -a $OUT/matrix.txt; echo | tee -a $OUT/matrix.txt
  echo "== gangs 20k x 200k ASCHED_HELPERS_LOCAL=$loc" | tee -a $OUT/matrix.txt
  ASCHED_HELPERS_LOCAL=$loc timeout 300 python tools/prof_config4.py gangs 2>&1 | tail -n 1 | tee -a $OUT/matrix.txt
  for ft in 1 0; do
    echo "== preempt 20k x 200k LOCAL=$loc FT=$ft" | tee -a $OUT/matrix.txt
    ASCHED_HELPERS_LOCAL=$loc ASCHED_FT=$ft timeout 300 python tools/prof_config4.py 2>&1 | tail -n 1 | tee -a $OUT/matrix.txt
    echo "== full-size configs[4] LOCAL=$loc FT=$ft" | tee -a $OUT/matrix.txt
    ASCHED_HELPERS_LOCAL=$loc ASCHED_FT=$ft timeout 600 python tools/prof_config4.py full 2>&1 | tail -n 1 | tee -a $OUT/matrix.txt
  done
done
for ft in 1 0; do
  echo "== segments (profiling build), preempt 20k x 200k, LOCAL=1 FT=$ft" | tee -a $OUT/segments.txt
  ASCHED_FT=$ft ASCHED_LIB_PATH=$PWD/armada_amd/csrc/libarmada_sched_prof.so ASCHED_PRINT_SEG=1 timeout 300 python tools/prof_config4.py 2>&1 | tail -n 2 | tee -a $OUT/segments.txt
done
