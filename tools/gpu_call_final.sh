#!/usr/bin/env bash
# The measurement call of a round: the driver-shaped bench line (headline + other_configs), ONE rocprofv3 kernel-stats CSV PER CONFIG (the round-2 review: a single CSV
# over the whole bench command mixes configs[2] / [3] / [4] launches of k_control), the PMC passes for roofline.traffic, the host-phase profile.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_call_final.sh r03x'
# Writes gpurun_out/<tag>/ ; copy what is worth keeping into profiles/.
set -u
TAG=${1:-r03x}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/summary.txt"
timeout 1500 python bench.py --steps 20 --warmup 5 > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"; echo "bench rc=$?" | tee -a "$OUT/summary.txt"
prof() {   # name, bench flags
  local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_$name" -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-budget 0 --no-other "$@" > "$OLDPWD/$OUT/prof_$name.log" 2>&1 )
  local DB; DB=$(find "$OUT/prof_$name" -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "$OUT/kernel_stats_$name.csv" > /dev/null
  find "$OUT/prof_$name" -name "*.db" -size +8M -delete
}
prof config2_headline
prof config3_gangs --gangs 10000
prof config4_reduced --nodes 20000 --jobs 200000 --queues 32 --occupied 0.95
prof config4_full --occupied 0.95 --steps 1 --warmup 0
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_config1" -- python -c "
import sys; sys.path.insert(0, '$OLDPWD'); sys.argv=['bench.py']
import torch; torch.cuda.init()   # torch's bundled HIP runtime before the library's (tests/conftest.py)
import bench, argparse, armada_amd
a = argparse.Namespace(other_scale=1.0, steps=10, cpu_budget=0)
print(bench.fit_batch_record(armada_amd.load_library(), a)['device_ms'])" > "$OLDPWD/$OUT/prof_config1.log" 2>&1 )
DB=$(find "$OUT/prof_config1" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "$OUT/kernel_stats_config1_fit_batch.csv" > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$OLDPWD/$OUT/pmc_$C" -- python "$OLDPWD/bench.py" --steps 1 --warmup 0 --cpu-budget 0 --no-other > "$OLDPWD/$OUT/pmc_$C.log" 2>&1 )
done
F=$(find "$OUT/pmc_FETCH_SIZE" -name "*.db" | head -1); W=$(find "$OUT/pmc_WRITE_SIZE" -name "*.db" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_summary.py "$F" "$W" "$OUT/pmc_hbm_traffic.json" > /dev/null
# the same two counters on the production-shaped configs[4] round (bench.py's checker record reads this file for its roofline.traffic)
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d "$OLDPWD/$OUT/pmc4_$C" -- python "$OLDPWD/tools/prof_config4.py" checker > "$OLDPWD/$OUT/pmc4_$C.log" 2>&1 )
done
F=$(find "$OUT/pmc4_FETCH_SIZE" -name "*.db" | head -1); W=$(find "$OUT/pmc4_WRITE_SIZE" -name "*.db" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_summary.py "$F" "$W" "$OUT/pmc_hbm_traffic_config4_checker.json" > /dev/null
find "$OUT" -name "*.db" -size +8M -delete
ASCHED_HOSTPROF=1 timeout 300 python bench.py --steps 1 --warmup 0 --cpu-budget 0 --no-other > "$OUT/bench_hostprof.json" 2> "$OUT/bench_hostprof.err"; grep hostprof "$OUT/bench_hostprof.err" | tail -20 > "$OUT/hostprof.txt"
head -c 900 "$OUT/bench_full.json" | tee -a "$OUT/summary.txt"; echo; ls "$OUT" | tee -a "$OUT/summary.txt"
timeout 1500 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "pytest(all gpu) rc=$?" | tee -a "$OUT/summary.txt"
tail -4 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
