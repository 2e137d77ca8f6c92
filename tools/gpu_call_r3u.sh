#!/usr/bin/env bash
# round 3, call u: where does a gang session (fastGangRun) spend its time?  profiling build, statSeg[26..31]
OUT=gpurun_out/${1:-r03u}; mkdir -p $OUT
for shape in gangsfull gangs; do
  echo "== $shape (profiling build)" | tee -a $OUT/summary.txt
  ASCHED_LIB_PATH=$PWD/armada_amd/csrc/libarmada_sched_prof.so ASCHED_PRINT_SEG=1 timeout 600 python tools/prof_config4.py $shape 2>&1 | grep -v amdgpu.ids | cut -c1-1500 | tee -a $OUT/summary.txt
done
