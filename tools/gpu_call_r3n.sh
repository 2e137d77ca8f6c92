#!/usr/bin/env bash
# round 3, call n: k_opt_score_wave (one wave per node) against the oracle and against the one-node-per-thread kernel
OUT=gpurun_out/${1:-r03n}; mkdir -p $OUT
export ASCHED_SAFETY_DEADLINE_S=120
timeout 900 python -m pytest tests/test_z_optimiser.py tests/test_z_optimiser_round.py tests/test_gpu_parity.py -q -m gpu -k "optimiser or scores or preempts or job_checks or private" > $OUT/pytest_opt.log 2>&1; echo "pytest(optimiser) rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_opt.log | tee -a $OUT/summary.txt
for mode in 0 1; do
  ASCHED_OPT_PER_THREAD=$mode python - <<'PY' 2>&1 | tee -a $OUT/summary.txt
import sys, argparse, os
sys.argv=['bench.py']
import torch; torch.cuda.init()   # torch's HIP runtime first (see tests/conftest.py)
import bench, armada_amd
a = argparse.Namespace(other_scale=1.0, steps=10, cpu_budget=30)
r = bench.optimiser_record(armada_amd.load_library(), a)
print("ASCHED_OPT_PER_THREAD=" + os.environ["ASCHED_OPT_PER_THREAD"], {k: r[k] for k in ("value", "host_ms_per_job", "k_opt_score_ms", "parity")})
PY
done
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_opt" -- python -c "
import sys; sys.path.insert(0, '$OLDPWD'); sys.argv=['bench.py']
import torch; torch.cuda.init()
import bench, argparse, armada_amd
a = argparse.Namespace(other_scale=1.0, steps=10, cpu_budget=0)
print(bench.optimiser_record(armada_amd.load_library(), a)['k_opt_score_ms'])" > "$OLDPWD/$OUT/prof_opt.log" 2>&1 )
DB=$(find "$OUT/prof_opt" -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "$OUT/kernel_stats_optimiser_scoring.csv" > /dev/null
find "$OUT" -name "*.db" -size +8M -delete
grep -i "opt_\|price" $OUT/kernel_stats_optimiser_scoring.csv | tee -a $OUT/summary.txt
