#!/usr/bin/env bash
# One consolidated GPU call for the start of a round (gpurun budget is 90 min; every call pays minutes of overhead):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_first_call.sh r02a'
# Writes everything under gpurun_out/<tag>/ ; copy the summaries worth keeping into profiles/ afterwards.
#   1. pytest -m gpu (no -x: the whole picture, incl. the 126 tests that have never run on a GPU)
#   2. bench.py default (rounds/s, p99, roofline, cpu_baseline, input_build_s)
#   3. rocprofv3 kernel stats of one bench round, then the two PMC passes (separate runs, as the pool requires)
#   4. submit-check throughput
set -u
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/summary.txt"
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.txt"
tail -5 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"; echo "bench rc=$?" | tee -a "$OUT/summary.txt"
ASCHED_HOSTPROF=1 timeout 600 python bench.py --steps 1 --warmup 0 --cpu-budget 0 > "$OUT/bench_hostprof.json" 2> "$OUT/bench_hostprof.err"
grep hostprof "$OUT/bench_hostprof.err" | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_stats" -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-budget 0 > "$OLDPWD/$OUT/prof_stats.log" 2>&1 )
DB=$(find "$OUT/prof_stats" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "$OUT/kernel_stats.csv" > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d "$OLDPWD/$OUT/pmc_$C" -- python "$OLDPWD/bench.py" --steps 1 --warmup 0 --cpu-budget 0 > "$OLDPWD/$OUT/pmc_$C.log" 2>&1 )
done
F=$(find "$OUT/pmc_FETCH_SIZE" -name "*.db" | head -1); W=$(find "$OUT/pmc_WRITE_SIZE" -name "*.db" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_summary.py "$F" "$W" "$OUT/pmc_hbm_traffic.json" > /dev/null
timeout 600 python bench.py --submit-check --steps 3 > "$OUT/bench_submitcheck.json" 2> "$OUT/bench_submitcheck.err"; echo "submitcheck bench rc=$?" | tee -a "$OUT/summary.txt"
# keep the merge-back small: the rocprof databases stay on the box
find "$OUT" -name "*.db" -size +8M -delete
cat "$OUT/bench_full.json" | head -c 600 | tee -a "$OUT/summary.txt"
