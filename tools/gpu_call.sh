#!/usr/bin/env bash
# One consolidated GPU call (gpurun budget is 90 min per round; every call pays minutes of overhead):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_call.sh r02a "at_scale" 10'
# args: <tag> <pytest -k expression, "" = all gpu tests, "none" = skip pytest> <bench steps> [extra bench flags]
# Writes everything under gpurun_out/<tag>/ ; copy the summaries worth keeping into profiles/ afterwards.
set -u
TAG=${1:-r02a}; KEXPR=${2:-}; STEPS=${3:-10}; EXTRA=${4:-}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/summary.txt"
if [ "$KEXPR" != "none" ]; then
  if [ -n "$KEXPR" ]; then timeout 1800 python -m pytest tests -q -m gpu -k "$KEXPR" -p no:cacheprovider --durations=15 > "$OUT/pytest_gpu.log" 2>&1
  else timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > "$OUT/pytest_gpu.log" 2>&1; fi
  echo "pytest rc=$?" | tee -a "$OUT/summary.txt"; tail -25 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
fi
timeout 1200 python bench.py --steps "$STEPS" --warmup 2 $EXTRA > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"; echo "bench rc=$?" | tee -a "$OUT/summary.txt"
# kernel stats of the same command (fewer steps, no oracle legs): k_control, k_fit_batch and k_control_aux rows in one table
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_stats" -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-budget 0 $EXTRA > "$OLDPWD/$OUT/prof_stats.log" 2>&1 )
DB=$(find "$OUT/prof_stats" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "$OUT/kernel_stats.csv" > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d "$OLDPWD/$OUT/pmc_$C" -- python "$OLDPWD/bench.py" --steps 1 --warmup 0 --cpu-budget 0 --no-other $EXTRA > "$OLDPWD/$OUT/pmc_$C.log" 2>&1 )
done
F=$(find "$OUT/pmc_FETCH_SIZE" -name "*.db" | head -1); W=$(find "$OUT/pmc_WRITE_SIZE" -name "*.db" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_summary.py "$F" "$W" "$OUT/pmc_hbm_traffic.json" > /dev/null
find "$OUT" -name "*.db" -size +8M -delete
head -c 1500 "$OUT/bench_full.json" | tee -a "$OUT/summary.txt"
ASCHED_HOSTPROF=1 timeout 300 python bench.py --steps 1 --warmup 0 --cpu-budget 0 --no-other > "$OUT/bench_hostprof.json" 2> "$OUT/bench_hostprof.err"; grep hostprof "$OUT/bench_hostprof.err" | tail -20 | tee -a "$OUT/summary.txt"
