#!/usr/bin/env bash
# round 3, call q: where does the gang-heavy round (configs[3]) spend its time?  engine-busy vs control-wait clocks of the profiling build, next to the headline shape
OUT=gpurun_out/${1:-r03q}; mkdir -p $OUT
export ASCHED_SAFETY_DEADLINE_S=120
for shape in headline gangsfull gangs; do
  echo "== $shape (profiling build)" | tee -a $OUT/summary.txt
  ASCHED_LIB_PATH=$PWD/armada_amd/csrc/libarmada_sched_prof.so ASCHED_PRINT_SEG=1 timeout 600 python tools/prof_config4.py $shape 2>&1 | grep -v amdgpu.ids | cut -c1-1500 | tee -a $OUT/summary.txt
done
