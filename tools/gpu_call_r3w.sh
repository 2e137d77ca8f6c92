#!/usr/bin/env bash
# 256 queues (generic iteration): segment profile
set -u
OUT=gpurun_out/r3w; mkdir -p "$OUT"
export TMPDIR=/tmp
ASCHED_LIB_PATH=armada_amd/csrc/libarmada_sched_prof.so ASCHED_PRINT_SEG=1 timeout 600 python - > "$OUT/q256_prof.log" 2>&1 <<'PY'
import sys, json, types; sys.path.insert(0, '.')
import torch; torch.cuda.init()
import armada_amd, bench
hip = armada_amd.load_library()
args = types.SimpleNamespace(other_scale=1.0, cpu_budget=0.0)
from armada_amd.binding import Scheduler
orig = Scheduler.round_stats
rec, wl, res, iters = bench.round_shape_record(hip, args, "256 queues", dict(n_nodes=20_000, n_jobs=200_000, n_queues=256), 1, "", 0)
print(json.dumps({"ms_per_step": rec["ms_per_step"], "round": rec["round"]}), flush=True)
PY
echo "rc=$?" | tee -a "$OUT/summary.txt"; cut -c1-1500 "$OUT/q256_prof.log" | tee -a "$OUT/summary.txt"
