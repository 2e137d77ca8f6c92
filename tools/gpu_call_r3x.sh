#!/usr/bin/env bash
# after r03v (k_control unchanged: tools/kcontrol_isa_hash.sh): the driver-shaped bench line with the new sub-records, the GPU suite, and the differential soaks
# through the PRODUCT library (SOAK_LIB=hip).   /usr/local/graft/bin/gpurun --timeout 2000 -- 'bash tools/gpu_call_r3x.sh'
set -u
OUT=gpurun_out/${TAG:-r03z}; mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/summary.txt"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"; echo "bench rc=$?" | tee -a "$OUT/summary.txt"
head -c 700 "$OUT/bench_full.json" | tee -a "$OUT/summary.txt"; echo
for k in "market 400" "rounds 400" "preempt 200" "features 100" "away 150" "optimiser 150" "offgrid 200" "streams 120"; do
  SOAK_LIB=hip timeout 400 python tests/soak.py $k 2>&1 | tail -1 | tee -a "$OUT/soak_hip.txt"
done
timeout 1200 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "pytest(all gpu) rc=$?" | tee -a "$OUT/summary.txt"
tail -4 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
