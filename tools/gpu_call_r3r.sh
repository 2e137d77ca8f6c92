#!/usr/bin/env bash
# round 3, call r: stream back-off policy (how soon the fast loop tries to start a stream run again after a short one) on the gang-heavy rounds and the headline, A/B in one call
OUT=gpurun_out/${1:-r03r}; mkdir -p $OUT
export ASCHED_SAFETY_DEADLINE_S=120
for rep in 1 2; do
  for lib in armada_amd/csrc/libarmada_sched.so armada_amd/csrc/libarmada_sched_bo64.so armada_amd/csrc/libarmada_sched_bo16.so armada_amd/csrc/libarmada_sched_bo1.so; do
    name=$(basename $lib .so); name=${name#libarmada_sched}; name=${name:-_default}
    for shape in gangs gangsfull headline; do
      echo "== $name $shape rep $rep" | tee -a $OUT/summary.txt
      ASCHED_LIB_PATH=$PWD/$lib timeout 600 python tools/prof_config4.py $shape 2>&1 | grep "^round" | cut -c1-420 | tee -a $OUT/summary.txt
    done
  done
done
