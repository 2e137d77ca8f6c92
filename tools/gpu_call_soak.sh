#!/usr/bin/env bash
# differential soaks through the PRODUCT library on the GPU box (tests/soak.py, SOAK_LIB=hip): many more seeds than the suite runs
#   TAG=r04z2 /usr/local/graft/bin/gpurun --timeout 2400 -- 'TAG=r04z2 bash tools/gpu_call_soak.sh'          (optional arguments: "kind seeds" pairs instead of the default list)
# Every kind runs under its own time limit (KIND_LIMIT seconds, default 420) with SOAK_PROGRESS=1: a round kernel that hangs costs that limit, not the whole call, and the
# summary names the seed it hung on (round 5 lost 25 GPU-minutes to one hung seed before this: profiles/r05y_bulk_skip_hang.txt).
set -u
OUT=gpurun_out/${TAG:-r04z2}; mkdir -p "$OUT"
export TMPDIR=/tmp
LIMIT=${KIND_LIMIT:-420}
if [ $# -gt 0 ]; then KINDS=("$@"); else KINDS=("preempt 1500" "rounds 2000" "streams 500" "market 1500" "optimiser 800" "away 800" "offgrid 800" "features 400" "ops 3000" "fit 400" "wide 300" "excluded 1500" "submitcheck 400"); fi
python -c "import torch; torch.cuda.init()" > /dev/null 2>&1   # (the first import of torch on a fresh box takes a minute or two: not inside a kind's limit)
for k in "${KINDS[@]}"; do   # "kind seeds [VAR=value ...]": the optional settings go into that kind's environment (e.g. "streams 1500 ASCHED_MERGE_MIN=64": small rounds through the bulk merge + split engine)
  set -- $k; KIND=$1; N=$2; shift 2; EXTRA="$*"
  LOG="$OUT/soak_${KIND}${EXTRA:+_$(echo "$EXTRA" | tr -c 'A-Za-z0-9=\n' '_')}.log"
  env $EXTRA SOAK_LIB=hip SOAK_PROGRESS=1 timeout "$LIMIT" python tests/soak.py $KIND $N > "$LOG" 2>&1; rc=$?
  if [ $rc = 124 ]; then echo "$k: TIME LIMIT ($LIMIT s) — last seed started: $(grep '^seed [0-9]*$' "$LOG" | tail -n 1)" | tee -a "$OUT/soak_hip.txt"
  else grep -v '^seed [0-9]*$' "$LOG" | tail -n 1 | tee -a "$OUT/soak_hip.txt"; fi
  grep -v '^seed [0-9]*$' "$LOG" | grep '^seed ' | head -n 5 >> "$OUT/soak_hip.txt"   # (divergences, if any)
done
