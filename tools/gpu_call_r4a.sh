#!/usr/bin/env bash
# round 4, first call: r03 library ("base") vs the working tree ("new": LANE0_PUBLISHED fences + helper result slots) in one box
set -u
bash tools/ab_call.sh r04a "base new" "headline gangs preempt" "at_scale or 64k or lds_idiom"
for L in base new; do
  P=$PWD/armada_amd/csrc/libarmada_sched_$L.so; [ $L = new ] && P=$PWD/armada_amd/csrc/libarmada_sched.so
  echo "== $L config4 full" >> gpurun_out/r04a/ab.txt
  ASCHED_LIB_PATH=$P timeout 400 python tools/prof_config4.py full 2>&1 | tail -n 1 >> gpurun_out/r04a/ab.txt
done
tail -n 4 gpurun_out/r04a/ab.txt
