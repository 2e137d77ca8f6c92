#!/usr/bin/env bash
# configs[4] (reduced and full): segment profile of the preempting iteration
set -u
OUT=gpurun_out/r3w; mkdir -p "$OUT"
export TMPDIR=/tmp
for shape in reduced full; do
  ASCHED_LIB_PATH=armada_amd/csrc/libarmada_sched_prof.so ASCHED_PRINT_SEG=1 timeout 400 python tools/prof_config4.py $shape > "$OUT/c4_$shape.log" 2>&1; echo "$shape rc=$?" | tee -a "$OUT/summary.txt"
  cut -c1-1600 "$OUT/c4_$shape.log" | tee -a "$OUT/summary.txt"
done
