#!/usr/bin/env bash
# A/B inside ONE GPU call (boxes differ by a few %): the library of the working tree against armada_amd/csrc/libarmada_sched_base.so
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/ab_call.sh r02t'
set -u
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
NEW=$PWD/armada_amd/csrc/libarmada_sched.so; BASE=$PWD/armada_amd/csrc/libarmada_sched_base.so
for rep in 1 2; do
  for L in base new; do
    P=$NEW; [ $L = base ] && P=$BASE
    echo "== $L gangs (rep $rep)" >> "$OUT/ab.txt"; ASCHED_LIB_PATH=$P timeout 300 python tools/prof_config4.py gangs >> "$OUT/ab.txt" 2>&1
  done
done
for L in base new; do
  P=$NEW; [ $L = base ] && P=$BASE
  echo "== $L headline" >> "$OUT/ab.txt"; ASCHED_LIB_PATH=$P timeout 600 python bench.py --steps 6 --warmup 2 --cpu-budget 0 --no-other 2>/dev/null | head -c 600 >> "$OUT/ab.txt"; echo >> "$OUT/ab.txt"
  echo "== $L preemption-heavy" >> "$OUT/ab.txt"; ASCHED_LIB_PATH=$P timeout 300 python tools/prof_config4.py >> "$OUT/ab.txt" 2>&1
done
timeout 900 python -m pytest tests -q -m gpu -k "stream or gang or at_scale" -p no:cacheprovider > "$OUT/pytest_ab.log" 2>&1; echo "pytest rc=$?" >> "$OUT/ab.txt"; tail -n 5 "$OUT/pytest_ab.log" >> "$OUT/ab.txt"
cat "$OUT/ab.txt"
