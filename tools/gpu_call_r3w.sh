#!/usr/bin/env bash
# rocprofv3 kernel stats of the final library: headline and configs[4] reduced (one CSV per config, like tools/gpu_call_final.sh)
set -u
TAG=r03z; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
prof() {
  local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_$name" -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-budget 0 --no-other "$@" > "$OLDPWD/$OUT/prof_$name.log" 2>&1 )
  local DB; DB=$(find "$OUT/prof_$name" -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "$OUT/kernel_stats_$name.csv" > /dev/null
  find "$OUT/prof_$name" -name "*.db" -size +8M -delete
  head -4 "$OUT/kernel_stats_$name.csv"
}
prof config2_headline
prof config4_reduced --nodes 20000 --jobs 200000 --queues 32 --occupied 0.95
