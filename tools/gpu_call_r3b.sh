#!/usr/bin/env bash
# round 3, call b: after the aliasing fix — minsize variants' fingerprints, A/B of -fno-strict-aliasing on the headline / gang / preempt shapes
set -u
OUT=gpurun_out/r3b; mkdir -p $OUT
for L in default selectAtPriority selectWithFairPreemption all; do
  P=$PWD/armada_amd/csrc/libarmada_sched_ms_$L.so; [ $L = default ] && P=$PWD/armada_amd/csrc/libarmada_sched.so
  echo "== $L" | tee -a $OUT/bisect_after_fix.txt
  ASCHED_LIB_PATH=$P timeout 300 python tools/round_fingerprint.py 2>&1 | tail -n 1 | tee -a $OUT/bisect_after_fix.txt
done
bash tools/ab_call.sh r3b "base new" "headline gangs preempt" "at_scale or order_key or fast_path"
