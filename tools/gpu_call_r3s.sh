#!/usr/bin/env bash
# round 3, call s: ASCHED_PREP_ONE (a queue's stream prepared by the control wave alone: after its gang, and whenever the queue at the top has none) A/B in one library
OUT=gpurun_out/${1:-r03s}; mkdir -p $OUT
export ASCHED_SAFETY_DEADLINE_S=120
for rep in 1 2; do
  for flag in 0 1; do
    for shape in gangs gangsfull headline; do
      echo "== prep_one=$flag $shape rep $rep" | tee -a $OUT/summary.txt
      ASCHED_PREP_ONE=$flag timeout 600 python tools/prof_config4.py $shape 2>&1 | grep "^round" | cut -c1-420 | tee -a $OUT/summary.txt
    done
  done
done
ASCHED_PREP_ONE=1 timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu_prep_one.log 2>&1; echo "pytest(all gpu, ASCHED_PREP_ONE=1) rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu_prep_one.log | tee -a $OUT/summary.txt
