#!/usr/bin/env bash
# differential soaks through the PRODUCT library on the GPU box (tests/soak.py, SOAK_LIB=hip): many more seeds than the suite runs
#   TAG=r04z2 /usr/local/graft/bin/gpurun --timeout 2400 -- 'TAG=r04z2 bash tools/gpu_call_soak.sh'          (optional arguments: "kind seeds" pairs instead of the default list)
set -u
OUT=gpurun_out/${TAG:-r04z2}; mkdir -p "$OUT"
export TMPDIR=/tmp
if [ $# -gt 0 ]; then KINDS=("$@"); else KINDS=("preempt 1500" "rounds 2000" "streams 500" "market 1500" "optimiser 800" "away 800" "offgrid 800" "features 400" "ops 3000" "fit 400" "wide 300" "excluded 1500" "submitcheck 400"); fi
for k in "${KINDS[@]}"; do
  SOAK_LIB=hip timeout 900 python tests/soak.py $k 2>&1 | tail -1 | tee -a "$OUT/soak_hip.txt"
done
